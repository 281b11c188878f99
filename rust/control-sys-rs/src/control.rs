pub mod batch;
pub mod controller;
mod device;
pub mod model;
pub mod simulator;
pub mod trajectory;
