// control_sys_rs.hpp -- C++17 host side of the MI355X batched MPC path, above the C ABI of include/mpcx.h.
//
// The reference crate (`control_sys_rs`, Rust + nalgebra) is compiled code; this header mirrors its public API
// one to one in C++ -- same module / type / function names, argument order and meaning, and error behaviour -- so
// that code written against the reference reads the same here:
//
//   control_sys_rs::control::model        DiscreteStateSpaceModel, dc_motor::{Parameters, Model},
//                                         mpc::{Parameters, Model}                       (src/control/model.rs)
//   control_sys_rs::control::controller   mpc::Controller                                (src/control/controller.rs)
//   control_sys_rs::control::simulator    compute_system_response, step                  (src/control/simulator.rs)
//   control_sys_rs::control::trajectory   generate_pulse_trajectory                      (src/control/trajectory.rs)
//   control_sys_rs::control::batch        additive: many plants per call (no counterpart in the reference)
//   na::DMatrix, na::DVector              the slice of nalgebra's dynamic f64 types the API shows: column-major,
//                                         contiguous, as_ptr()/as_slice() are what crosses the C ABI
//
// Every computation of the hot path (plant discretisation, lifted matrices, Hessian, inversion + gain, control move,
// propagation, rollout) is a call into libmpcx.so and runs on the GPU.  There is no host arithmetic fallback: without
// a usable device the first call throws DeviceError.
//
// Rust -> C++ conventions used here
//   Result<T, Box<dyn Error>>::Err(msg)   -> throws control_sys_rs::Error(msg)          (e.g. "Matrix inversion failed")
//   panic!(..) / unwrap() on None         -> throws control_sys_rs::Panic(msg)          (shape mismatch, slicing, unwrap)
//   Rc<dyn DiscreteStateSpaceModel>       -> std::shared_ptr<const DiscreteStateSpaceModel>
//   Type::new(..)                         -> Type::new_(..) (and a constructor where the type is constructible)
//   &mut self / &self                     -> non-const / const member functions
//   A Controller is bound to the thread that made it (the reference's holds an Rc and is !Send): the device context
//   is thread_local, one per host thread.
#ifndef CONTROL_SYS_RS_HPP
#define CONTROL_SYS_RS_HPP

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "mpcx.h"

namespace control_sys_rs {

struct Error : std::runtime_error {  // the Err(..) arm of a Result
    using std::runtime_error::runtime_error;
};
struct Panic : std::logic_error {  // what would be a panic in the reference
    using std::logic_error::logic_error;
};
struct DeviceError : std::runtime_error {  // the GPU library refused or failed; never masked by host arithmetic
    int code;
    DeviceError(int c, const std::string &what) : std::runtime_error(what), code(c) {}
};

}  // namespace control_sys_rs

namespace na {

// Dynamic f64 matrix, column-major and contiguous like nalgebra's DMatrix<f64> (VecStorage).
class DMatrix {
  public:
    DMatrix() = default;
    DMatrix(std::size_t nrows, std::size_t ncols, double fill = 0.0) : r_(nrows), c_(ncols), d_(nrows * ncols, fill) {}
    static DMatrix zeros(std::size_t nrows, std::size_t ncols) { return DMatrix(nrows, ncols, 0.0); }
    static DMatrix from_element(std::size_t nrows, std::size_t ncols, double x) { return DMatrix(nrows, ncols, x); }
    static DMatrix identity(std::size_t nrows, std::size_t ncols) {
        DMatrix a(nrows, ncols, 0.0);
        for (std::size_t i = 0; i < nrows && i < ncols; ++i) a(i, i) = 1.0;
        return a;
    }
    // rows given one after the other, as the dmatrix![..] macro reads
    static DMatrix from_row_slice(std::size_t nrows, std::size_t ncols, const std::vector<double> &rows) {
        if (rows.size() != nrows * ncols) throw control_sys_rs::Panic("Matrix init. error: the slice did not contain the right number of elements.");
        DMatrix a(nrows, ncols);
        for (std::size_t i = 0; i < nrows; ++i)
            for (std::size_t j = 0; j < ncols; ++j) a(i, j) = rows[i * ncols + j];
        return a;
    }
    std::size_t nrows() const { return r_; }
    std::size_t ncols() const { return c_; }
    std::pair<std::size_t, std::size_t> shape() const { return {r_, c_}; }
    double &operator()(std::size_t i, std::size_t j) { return d_[check(i, j)]; }
    double operator()(std::size_t i, std::size_t j) const { return d_[check(i, j)]; }
    double *as_mut_ptr() { return d_.data(); }
    const double *as_ptr() const { return d_.data(); }
    const std::vector<double> &as_slice() const { return d_; }
    DMatrix column(std::size_t j) const {
        DMatrix v(r_, 1);
        for (std::size_t i = 0; i < r_; ++i) v(i, 0) = (*this)(i, j);
        return v;
    }
    std::vector<double> row(std::size_t i) const {
        std::vector<double> v(c_);
        for (std::size_t j = 0; j < c_; ++j) v[j] = (*this)(i, j);
        return v;
    }

  private:
    std::size_t check(std::size_t i, std::size_t j) const {
        if (i >= r_ || j >= c_) throw control_sys_rs::Panic("Matrix index out of bounds.");
        return j * r_ + i;
    }
    std::size_t r_ = 0, c_ = 0;
    std::vector<double> d_;
};

// Dynamic f64 column vector (nalgebra's DVector<f64>).
class DVector {
  public:
    DVector() = default;
    explicit DVector(std::size_t n, double fill = 0.0) : d_(n, fill) {}
    explicit DVector(std::vector<double> v) : d_(std::move(v)) {}
    static DVector zeros(std::size_t n) { return DVector(n, 0.0); }
    static DVector from_element(std::size_t n, double x) { return DVector(n, x); }
    std::size_t nrows() const { return d_.size(); }
    std::size_t len() const { return d_.size(); }
    double &operator[](std::size_t i) { return d_[check(i)]; }
    double operator[](std::size_t i) const { return d_[check(i)]; }
    double *as_mut_ptr() { return d_.data(); }
    const double *as_ptr() const { return d_.data(); }
    const std::vector<double> &as_slice() const { return d_; }

  private:
    std::size_t check(std::size_t i) const {
        if (i >= d_.size()) throw control_sys_rs::Panic("Matrix index out of bounds.");
        return i;
    }
    std::vector<double> d_;
};

}  // namespace na

namespace control_sys_rs {
namespace detail {

// One device context per host thread, created on first use on device MPCX_DEVICE (default 0).
class Device {
  public:
    static mpcx_ctx *ctx() {
        thread_local Device dev;
        return dev.ctx_;
    }
    static void check(int rc, const char *what) {
        if (rc == MPCX_OK) return;
        const char *msg = mpcx_last_error(ctx());
        throw DeviceError(rc, std::string(what) + ": " + (msg && *msg ? msg : "error") + " (code " + std::to_string(rc) + ")");
    }

  private:
    Device() {
        const char *e = std::getenv("MPCX_DEVICE");
        const int id = e ? std::atoi(e) : 0;
        const int rc = mpcx_create(&ctx_, id);
        if (rc != MPCX_OK) {
            ctx_ = nullptr;
            throw DeviceError(rc, "mpcx_create(device " + std::to_string(id) + ") failed with code " + std::to_string(rc) +
                                      ": this library computes on the GPU only");
        }
    }
    ~Device() {
        if (ctx_) mpcx_destroy(ctx_);
    }
    mpcx_ctx *ctx_ = nullptr;
};

inline mpcx_dims dims(std::size_t n, std::size_t m, std::size_t r, std::size_t f, std::size_t v, std::int64_t batch = 1,
                      std::int64_t T = 0) {
    mpcx_dims d{};
    d.n = (std::int32_t)n; d.m = (std::int32_t)m; d.r = (std::int32_t)r; d.f = (std::int32_t)f; d.v = (std::int32_t)v;
    d.weights_mode = MPCX_WEIGHTS_BLOCK; d.hessian_algo = MPCX_HESSIAN_AUTO; d.u_shared = 0;
    d.batch = batch; d.T = T; d.interleave = MPCX_INTERLEAVE_AOS;
    return d;
}

// W4 as Controller::new builds it is block-diagonal with one repeated r x r block: true when `w4` has that form.
inline bool is_repeated_block(const na::DMatrix &w4, std::size_t f, std::size_t r) {
    for (std::size_t j = 0; j < f * r; ++j)
        for (std::size_t i = 0; i < f * r; ++i) {
            const double want = (i / r == j / r) ? w4(i % r, j % r) : 0.0;
            if (!(w4(i, j) == want)) return false;
        }
    return true;
}

}  // namespace detail

namespace control {

// ------------------------------------------------------------------------------------------------ model.rs
namespace model {

// model.rs:3-8
class DiscreteStateSpaceModel {
  public:
    virtual ~DiscreteStateSpaceModel() = default;
    virtual const na::DMatrix &get_mat_a() const = 0;
    virtual const na::DMatrix &get_mat_b() const = 0;
    virtual const na::DMatrix &get_mat_c() const = 0;
    virtual double get_sampling_dt() const = 0;
};

// Backward-Euler discretisation of a continuous plant on the device: A = (I - Ac dt)^-1, B = A (Bc dt)
// (model.rs:55-57, :138-140).  The reference unwraps the failed inversion: a singular I - Ac dt panics.
inline std::pair<na::DMatrix, na::DMatrix> c2d_backward_euler(const na::DMatrix &mat_ac, const na::DMatrix &mat_bc, double dt) {
    const std::size_t n = mat_ac.nrows(), m = mat_bc.ncols();
    if (mat_ac.ncols() != n) throw Panic("Unable to invert a non-square matrix.");
    if (mat_bc.nrows() != n) throw Panic("Matrix multiplication dimensions mismatch.");
    na::DMatrix a(n, n), b(n, m);
    std::int32_t info = -1;
    const mpcx_dims d = detail::dims(n, m, 1, 1, 1);
    detail::Device::check(mpcx_c2d_backward_euler_f64(detail::Device::ctx(), &d, mat_ac.as_ptr(), mat_bc.as_ptr(), dt,
                                                      a.as_mut_ptr(), b.as_mut_ptr(), &info), "mpcx_c2d_backward_euler_f64");
    if (info != 0) throw Panic("called `Option::unwrap()` on a `None` value");
    return {std::move(a), std::move(b)};
}

namespace detail_model {
class PlantModel : public DiscreteStateSpaceModel {
  public:
    const na::DMatrix &get_mat_a() const override { return mat_a_; }
    const na::DMatrix &get_mat_b() const override { return mat_b_; }
    const na::DMatrix &get_mat_c() const override { return mat_c_; }
    double get_sampling_dt() const override { return sampling_dt_; }

  protected:
    // Model::new on the device: continuous matrices from the parameters, then backward Euler (mpcx_plants_f64)
    PlantModel(int kind, std::size_t n, const double *params, double sampling_dt)
        : mat_a_(n, n), mat_b_(n, 1), mat_c_(1, n), sampling_dt_(sampling_dt) {
        std::int32_t info = -1;
        control_sys_rs::detail::Device::check(
            mpcx_plants_f64(control_sys_rs::detail::Device::ctx(), kind, 1, MPCX_INTERLEAVE_AOS, params, sampling_dt,
                            mat_a_.as_mut_ptr(), mat_b_.as_mut_ptr(), mat_c_.as_mut_ptr(), &info), "mpcx_plants_f64");
        if (info != 0) throw Panic("called `Option::unwrap()` on a `None` value");  // model.rs:56, :139
    }
    na::DMatrix mat_a_, mat_b_, mat_c_;
    double sampling_dt_;
};
}  // namespace detail_model

namespace dc_motor {
// model.rs:16-34.  The reference keeps the fields private and offers only the defaults; `with` is an addition.
class Parameters {
  public:
    Parameters() = default;  // Default::default(), model.rs:24-33
    static Parameters with(double b, double j, double k, double l, double r) {
        Parameters p;
        p.b = b; p.j = j; p.k = k; p.l = l; p.r = r;
        return p;
    }
    double b = 0.1, j = 0.01, k = 0.01, l = 0.5, r = 1.0;
};
// model.rs:36-86
class Model : public detail_model::PlantModel {
  public:
    Model(const Parameters &params, double sampling_dt) : PlantModel(MPCX_PLANT_DC_MOTOR, 2, pack(params).v, sampling_dt) {}
    static Model new_(const Parameters &params, double sampling_dt) { return Model(params, sampling_dt); }

  private:
    struct Packed { double v[5]; };
    static Packed pack(const Parameters &p) { return Packed{{p.b, p.j, p.k, p.l, p.r}}; }
};
}  // namespace dc_motor

namespace mpc {
// model.rs:95-115
class Parameters {
  public:
    Parameters() = default;  // Default::default(), model.rs:104-115
    static Parameters with(double m1, double m2, double k1, double k2, double d1, double d2) {
        Parameters p;
        p.m1 = m1; p.m2 = m2; p.k1 = k1; p.k2 = k2; p.d1 = d1; p.d2 = d2;
        return p;
    }
    double m1 = 2.0, m2 = 2.0, k1 = 100.0, k2 = 200.0, d1 = 1.0, d2 = 5.0;
};
// model.rs:117-169
class Model : public detail_model::PlantModel {
  public:
    Model(const Parameters &params, double sampling_dt) : PlantModel(MPCX_PLANT_TWO_MASS, 4, pack(params).v, sampling_dt) {}
    static Model new_(const Parameters &params, double sampling_dt) { return Model(params, sampling_dt); }

  private:
    struct Packed { double v[6]; };
    static Packed pack(const Parameters &p) { return Packed{{p.m1, p.m2, p.k1, p.k2, p.d1, p.d2}}; }
};
}  // namespace mpc

// A plant given directly by its discrete matrices (addition: the reference only has the two parametrised plants).
class Matrices : public DiscreteStateSpaceModel {
  public:
    Matrices(na::DMatrix a, na::DMatrix b, na::DMatrix c, double dt)
        : mat_a_(std::move(a)), mat_b_(std::move(b)), mat_c_(std::move(c)), sampling_dt_(dt) {}
    const na::DMatrix &get_mat_a() const override { return mat_a_; }
    const na::DMatrix &get_mat_b() const override { return mat_b_; }
    const na::DMatrix &get_mat_c() const override { return mat_c_; }
    double get_sampling_dt() const override { return sampling_dt_; }

  private:
    na::DMatrix mat_a_, mat_b_, mat_c_;
    double sampling_dt_;
};

}  // namespace model

// ------------------------------------------------------------------------------------------------ simulator.rs
namespace simulator {

namespace detail_sim {
// simulator.rs:5-35 (private there): returns (mat_y r x T, mat_x n x (T+1)) -- in that order, as the reference.
inline std::pair<na::DMatrix, na::DMatrix> system_simulate(const model::DiscreteStateSpaceModel &model, const na::DMatrix &mat_u,
                                                           const na::DVector &x0) {
    const na::DMatrix &a = model.get_mat_a(), &b = model.get_mat_b(), &c = model.get_mat_c();
    const std::size_t n = a.nrows(), m = b.ncols(), r = c.nrows(), T = mat_u.ncols();
    na::DMatrix mat_x = na::DMatrix::zeros(n, T + 1), mat_y = na::DMatrix::zeros(r, T);
    if (T == 0) return {std::move(mat_y), std::move(mat_x)};  // the loop body never runs: both stay zero (simulator.rs:13-15)
    if (mat_u.nrows() != m || x0.nrows() != n || a.ncols() != n || b.nrows() != n || c.ncols() != n)
        throw Panic("Matrix multiplication dimensions mismatch.");  // nalgebra's assertion at simulator.rs:18-21
    const mpcx_dims d = control_sys_rs::detail::dims(n, m, r, 1, 1, 1, (std::int64_t)T);
    control_sys_rs::detail::Device::check(
        mpcx_rollout_f64(control_sys_rs::detail::Device::ctx(), &d, a.as_ptr(), b.as_ptr(), c.as_ptr(), x0.as_ptr(),
                         mat_u.as_ptr(), mat_x.as_mut_ptr(), mat_y.as_mut_ptr()), "mpcx_rollout_f64");
    return {std::move(mat_y), std::move(mat_x)};
}
}  // namespace detail_sim

// simulator.rs:37-48: row 0 of the outputs
inline std::vector<double> compute_system_response(const model::DiscreteStateSpaceModel &model, const na::DMatrix &input,
                                                   const na::DVector &initial_state) {
    auto yx = detail_sim::system_simulate(model, input, initial_state);
    if (yx.first.nrows() == 0) throw Panic("Matrix index out of bounds.");
    return yx.first.row(0);
}

// simulator.rs:50-63: zero initial state, floor(duration / dt) samples of a unit input; (response, input, states)
inline std::tuple<na::DMatrix, na::DMatrix, na::DMatrix> step(const model::DiscreteStateSpaceModel &model, double duration) {
    const na::DVector initial_state = na::DVector::zeros(model.get_mat_a().nrows());
    const double q = std::floor(duration / model.get_sampling_dt());
    const std::size_t n_samples = q > 0.0 ? (std::size_t)q : 0;  // `as usize` saturates at 0
    na::DMatrix input = na::DMatrix::from_element(1, n_samples, 1.0);
    auto yx = detail_sim::system_simulate(model, input, initial_state);
    return {std::move(yx.first), std::move(input), std::move(yx.second)};
}

}  // namespace simulator

// ------------------------------------------------------------------------------------------------ trajectory.rs
namespace trajectory {
// trajectory.rs:3-11: ones on [0, T/3) and [2T/3, T); the second copy_from panics unless T - 2T/3 == T/3.
inline na::DMatrix generate_pulse_trajectory(std::size_t time_steps) {
    na::DMatrix traj = na::DMatrix::zeros(time_steps, 1);
    const std::size_t third = time_steps / 3, start = 2 * time_steps / 3;
    for (std::size_t i = 0; i < third; ++i) traj(i, 0) = 1.0;
    if (time_steps - start != third) throw Panic("Unable to copy from a matrix with incompatible shape.");
    for (std::size_t i = start; i < time_steps; ++i) traj(i, 0) = 1.0;
    return traj;
}
}  // namespace trajectory

// ------------------------------------------------------------------------------------------------ controller.rs
namespace controller {
namespace mpc {

class Controller {
  public:
    na::DMatrix desired_ctrl_traj_total;  // pub field, controller.rs:12

    // controller.rs:16-99: (mat_o, mat_m, gain_matrix); Err("Matrix inversion failed") -> throws Error.
    // Phi, Gamma, H = M^T W4 M + W3 come from mpcx_qp_build_f64, the inversion and H^-1 M^T W4 from mpcx_qp_gain_f64
    // (LU with partial pivoting, status per instance).
    static std::tuple<na::DMatrix, na::DMatrix, na::DMatrix> form_lifted_matrices(const na::DMatrix &mat_a, const na::DMatrix &mat_b,
                                                                                   const na::DMatrix &mat_c, std::size_t f,
                                                                                   std::size_t v, const na::DMatrix &mat_w3,
                                                                                   const na::DMatrix &mat_w4) {
        na::DMatrix mat_o, mat_m, h;
        mpcx_dims d;
        lifted_and_hessian(mat_a, mat_b, mat_c, f, v, mat_w3, mat_w4, mat_o, mat_m, h, d);
        const std::size_t r = mat_c.nrows(), m = mat_b.ncols();
        na::DMatrix gain(v * m, f * r);
        const na::DMatrix q = block_of(mat_w4, r);
        const double *w4 = d.weights_mode == MPCX_WEIGHTS_BLOCK ? q.as_ptr() : mat_w4.as_ptr();
        std::int32_t info = -1;
        control_sys_rs::detail::Device::check(
            mpcx_qp_gain_f64(control_sys_rs::detail::Device::ctx(), &d, h.as_ptr(), mat_m.as_ptr(), w4, gain.as_mut_ptr(), &info),
            "mpcx_qp_gain_f64");
        if (info != 0) throw Error("Matrix inversion failed");  // controller.rs:93-95
        return {std::move(mat_o), std::move(mat_m), std::move(gain)};
    }

    // Addition: (mat_o, rows) with rows = the first m rows of gain_matrix (m x f r), all that the control move uses
    // (controller.rs:130-132): m solves with H^T instead of f r solves (mpcx_qp_gain_rows_f64).  Throws as above.
    static std::pair<na::DMatrix, na::DMatrix> form_first_move_rows(const na::DMatrix &mat_a, const na::DMatrix &mat_b,
                                                                    const na::DMatrix &mat_c, std::size_t f, std::size_t v,
                                                                    const na::DMatrix &mat_w3, const na::DMatrix &mat_w4) {
        na::DMatrix mat_o, mat_m, h;
        mpcx_dims d;
        lifted_and_hessian(mat_a, mat_b, mat_c, f, v, mat_w3, mat_w4, mat_o, mat_m, h, d);
        const std::size_t r = mat_c.nrows(), m = mat_b.ncols();
        na::DMatrix rows(m, f * r);
        const na::DMatrix q = block_of(mat_w4, r);
        const double *w4 = d.weights_mode == MPCX_WEIGHTS_BLOCK ? q.as_ptr() : mat_w4.as_ptr();
        std::int32_t info = -1;
        control_sys_rs::detail::Device::check(
            mpcx_qp_gain_rows_f64(control_sys_rs::detail::Device::ctx(), &d, h.as_ptr(), mat_m.as_ptr(), w4, rows.as_mut_ptr(), &info),
            "mpcx_qp_gain_rows_f64");
        if (info != 0) throw Error("Matrix inversion failed");  // controller.rs:93-95
        return {std::move(mat_o), std::move(rows)};
    }

    // Addition: the matrix the reference hands to try_inverse, H = M^T W4 M + W3 (controller.rs:92-93), with O and M.
    static std::tuple<na::DMatrix, na::DMatrix, na::DMatrix> form_lifted_matrices_and_hessian(
        const na::DMatrix &mat_a, const na::DMatrix &mat_b, const na::DMatrix &mat_c, std::size_t f, std::size_t v,
        const na::DMatrix &mat_w3, const na::DMatrix &mat_w4) {
        na::DMatrix mat_o, mat_m, h;
        mpcx_dims d;
        lifted_and_hessian(mat_a, mat_b, mat_c, f, v, mat_w3, mat_w4, mat_o, mat_m, h, d);
        return {std::move(mat_o), std::move(mat_m), std::move(h)};
    }

    // controller.rs:137-207.  The weights are built on the host exactly as the reference builds them: every element of a
    // weight block is set to the weight (:164-165, :184).  `all_elements = false` (an addition) builds q I blocks, which
    // a plant with several inputs or outputs needs for the Hessian to be invertible.  `first_move_only = true` (an
    // addition) keeps only the first m rows of the gain matrix: no method of the controller reads more than those.
    static Controller new_(std::shared_ptr<const model::DiscreteStateSpaceModel> model, std::size_t f, std::size_t v, double q0,
                           double q_other, double pred_weight, const na::DMatrix &desired_ctrl_traj, bool all_elements = true,
                           bool first_move_only = false) {
        const std::size_t m = model->get_mat_b().ncols(), r = model->get_mat_c().nrows();
        const std::size_t vm = v * m;
        // W1 (:149-161) and W2 (:164-175)
        na::DMatrix w1 = na::DMatrix::zeros(vm, vm), w2 = na::DMatrix::zeros(vm, vm);
        for (std::size_t i = 0; i < v; ++i)
            for (std::size_t a = 0; a < m; ++a) {
                w1(i * m + a, i * m + a) = 1.0;
                if (i >= 1) w1(i * m + a, (i - 1) * m + a) = -1.0;
                for (std::size_t b = 0; b < m; ++b)
                    w2(i * m + a, i * m + b) = (all_elements || a == b) ? (i == 0 ? q0 : q_other) : 0.0;
            }
        // W3 = W1^T (W2 W1) (:177): host-side, shared by every plant, k-ascending multiply-add
        na::DMatrix w21 = na::DMatrix::zeros(vm, vm), w3 = na::DMatrix::zeros(vm, vm);
        for (std::size_t j = 0; j < vm; ++j)
            for (std::size_t k = 0; k < vm; ++k) {
                const double x = w1(k, j);
                if (x == 0.0) continue;
                for (std::size_t i = 0; i < vm; ++i) w21(i, j) += w2(i, k) * x;
            }
        for (std::size_t j = 0; j < vm; ++j)
            for (std::size_t i = 0; i < vm; ++i) {
                double s = 0.0;
                for (std::size_t k = 0; k < vm; ++k) s += w1(k, i) * w21(k, j);
                w3(i, j) = s;
            }
        // W4 (:179-185)
        na::DMatrix w4 = na::DMatrix::zeros(f * r, f * r);
        for (std::size_t i = 0; i < f; ++i)
            for (std::size_t a = 0; a < r; ++a)
                for (std::size_t b = 0; b < r; ++b) w4(i * r + a, i * r + b) = (all_elements || a == b) ? pred_weight : 0.0;
        Controller c;
        c.f_ = f;
        c.desired_ctrl_traj_total = desired_ctrl_traj;
        if (first_move_only) {
            auto orows = form_first_move_rows(model->get_mat_a(), model->get_mat_b(), model->get_mat_c(), f, v, w3, w4);
            c.v_ = 1;  // the stored gain is (1 m) x (f r)
            c.mat_o_ = std::move(orows.first);
            c.gain_matrix_ = std::move(orows.second);
        } else {
            auto omg = form_lifted_matrices(model->get_mat_a(), model->get_mat_b(), model->get_mat_c(), f, v, w3, w4);  // :190-198
            c.v_ = v;
            c.mat_o_ = std::move(std::get<0>(omg));
            c.gain_matrix_ = std::move(std::get<2>(omg));
        }
        c.model_ = std::move(model);
        return c;
    }

    // controller.rs:101-115: (x_kp1, y_k) with y_k = C x_k, the output at the current state.  One rollout step.
    std::pair<na::DVector, na::DVector> propagate_dynamics(const na::DVector &control_input, const na::DVector &state) const {
        const na::DMatrix &a = model_->get_mat_a(), &b = model_->get_mat_b(), &c = model_->get_mat_c();
        const std::size_t n = a.nrows(), m = b.ncols(), r = c.nrows();
        if (state.nrows() != n || control_input.nrows() != m) throw Panic("Matrix multiplication dimensions mismatch.");
        std::vector<double> x(2 * n);
        na::DVector y_k = na::DVector::zeros(r);
        const mpcx_dims d = control_sys_rs::detail::dims(n, m, r, 1, 1, 1, 1);
        control_sys_rs::detail::Device::check(
            mpcx_rollout_f64(control_sys_rs::detail::Device::ctx(), &d, a.as_ptr(), b.as_ptr(), c.as_ptr(), state.as_ptr(),
                             control_input.as_ptr(), x.data(), y_k.as_mut_ptr()), "mpcx_rollout_f64");
        return {na::DVector(std::vector<double>(x.begin() + (std::ptrdiff_t)n, x.end())), std::move(y_k)};
    }

    // controller.rs:117-135: first element of gain (traj[t .. t+f] - O x) as a 1-vector.  As the reference, one input and
    // one output (it takes element (0, 0) and slices a T x 1 trajectory); `compute_control_input_mimo` lifts that.
    na::DVector compute_control_input(std::size_t timestep, const na::DVector &state) {
        if (desired_ctrl_traj_total.ncols() != 1 || mat_o_.nrows() != f_) throw Panic("Matrix addition/subtraction dimensions mismatch.");
        const std::vector<double> u = first_move(timestep, state);
        na::DVector input_applied = na::DVector::zeros(1);
        input_applied[0] = u[0];
        return input_applied;
    }

    // Addition: the move for m inputs and r outputs -- the first m entries of gain (window - O x), the window being rows
    // timestep .. timestep + f of a (time_steps x r) trajectory, stacked time-major.
    na::DVector compute_control_input_mimo(std::size_t timestep, const na::DVector &state) { return na::DVector(first_move(timestep, state)); }

    // Addition: the loop of src/main.rs:51-71 in one device call: (states n x (steps+1), inputs m x steps,
    // outputs r x (steps+1)) with outputs[:, 0] = C x0 as main.rs:40-41 records it.
    std::tuple<na::DMatrix, na::DMatrix, na::DMatrix> closed_loop(std::size_t steps, const na::DVector &x0) const {
        const na::DMatrix &a = model_->get_mat_a(), &b = model_->get_mat_b(), &c = model_->get_mat_c();
        const std::size_t n = a.nrows(), m = b.ncols(), r = c.nrows();
        if (x0.nrows() != n) throw Panic("Matrix multiplication dimensions mismatch.");
        if (steps + f_ > desired_ctrl_traj_total.nrows() || desired_ctrl_traj_total.ncols() != r) throw Panic("Matrix slicing out of bounds.");
        na::DMatrix ref(r, steps + f_);  // the device takes r x (T + f), time along columns
        for (std::size_t t = 0; t < steps + f_; ++t)
            for (std::size_t q = 0; q < r; ++q) ref(q, t) = desired_ctrl_traj_total(t, q);
        na::DMatrix states(n, steps + 1), inputs(m, steps), y(r, steps), outputs(r, steps + 1);
        mpcx_dims d = control_sys_rs::detail::dims(n, m, r, f_, v_, 1, (std::int64_t)steps);
        d.u_shared = 1;
        control_sys_rs::detail::Device::check(
            mpcx_closed_loop_f64(control_sys_rs::detail::Device::ctx(), &d, a.as_ptr(), b.as_ptr(), c.as_ptr(), x0.as_ptr(),
                                 mat_o_.as_ptr(), gain_matrix_.as_ptr(), ref.as_ptr(), states.as_mut_ptr(), inputs.as_mut_ptr(),
                                 y.as_mut_ptr()), "mpcx_closed_loop_f64");
        // outputs[:, 0] = C x0 is Y[:, 0] of the loop; outputs[:, t+1] = y_t (main.rs:41, :63)
        for (std::size_t q = 0; q < r; ++q) {
            outputs(q, 0) = steps ? y(q, 0) : 0.0;
            for (std::size_t t = 0; t < steps; ++t) outputs(q, t + 1) = y(q, t);
        }
        if (steps == 0) {  // no step ran: C x0 through one propagate call
            auto xy = propagate_dynamics(na::DVector::zeros(m), x0);
            for (std::size_t q = 0; q < r; ++q) outputs(q, 0) = xy.second[q];
        }
        return {std::move(states), std::move(inputs), std::move(outputs)};
    }

    const na::DMatrix &mat_o() const { return mat_o_; }              // private in the reference; read-only here
    const na::DMatrix &gain_matrix() const { return gain_matrix_; }  // likewise

  private:
    Controller() = default;

    static na::DMatrix block_of(const na::DMatrix &w4, std::size_t r) {
        na::DMatrix q(r, r);
        for (std::size_t j = 0; j < r; ++j)
            for (std::size_t i = 0; i < r; ++i) q(i, j) = w4(i, j);
        return q;
    }

    static void lifted_and_hessian(const na::DMatrix &mat_a, const na::DMatrix &mat_b, const na::DMatrix &mat_c, std::size_t f,
                                   std::size_t v, const na::DMatrix &mat_w3, const na::DMatrix &mat_w4, na::DMatrix &mat_o,
                                   na::DMatrix &mat_m, na::DMatrix &h, mpcx_dims &d) {
        const std::size_t n = mat_a.nrows(), r = mat_c.nrows(), m = mat_b.ncols();  // controller.rs:28-30
        if (mat_a.ncols() != n || mat_b.nrows() != n || mat_c.ncols() != n)
            throw Panic("Matrix multiplication dimensions mismatch.");  // nalgebra's assertion at :38 / :43 / :61
        if (mat_w4.nrows() != f * r || mat_w4.ncols() != f * r) throw Panic("Matrix multiplication dimensions mismatch.");  // :92
        if (mat_w3.nrows() != v * m || mat_w3.ncols() != v * m) throw Panic("Matrix addition/subtraction dimensions mismatch.");  // :93
        d = control_sys_rs::detail::dims(n, m, r, f, v);
        const bool block = control_sys_rs::detail::is_repeated_block(mat_w4, f, r);
        d.weights_mode = block ? MPCX_WEIGHTS_BLOCK : MPCX_WEIGHTS_DENSE;
        const na::DMatrix q = block_of(mat_w4, r);
        mat_o = na::DMatrix::zeros(f * r, n);
        mat_m = na::DMatrix::zeros(f * r, v * m);
        h = na::DMatrix::zeros(v * m, v * m);
        control_sys_rs::detail::Device::check(
            mpcx_qp_build_f64(control_sys_rs::detail::Device::ctx(), &d, mat_a.as_ptr(), mat_b.as_ptr(), mat_c.as_ptr(),
                              mat_w3.as_ptr(), block ? q.as_ptr() : mat_w4.as_ptr(), nullptr, nullptr, mat_o.as_mut_ptr(),
                              mat_m.as_mut_ptr(), h.as_mut_ptr(), nullptr), "mpcx_qp_build_f64");
    }

    // gain (window - O x)[0 .. m) on the device: one step of the receding-horizon loop started at `state`
    // (mpcx_closed_loop_f64 with T = 1 keeps the reference's order: s first, then gain s).
    std::vector<double> first_move(std::size_t timestep, const na::DVector &state) const {
        const na::DMatrix &a = model_->get_mat_a(), &b = model_->get_mat_b(), &c = model_->get_mat_c();
        const std::size_t n = a.nrows(), m = b.ncols(), r = c.nrows();
        if (timestep + f_ > desired_ctrl_traj_total.nrows()) throw Panic("Matrix slicing out of bounds.");  // view_range, :121
        if (desired_ctrl_traj_total.ncols() != r) throw Panic("Matrix addition/subtraction dimensions mismatch.");  // :125
        if (state.nrows() != n) throw Panic("Matrix multiplication dimensions mismatch.");  // :126
        na::DMatrix ref = na::DMatrix::zeros(r, f_ + 1);  // r x (T + f) with T = 1; the last column is never read
        for (std::size_t i = 0; i < f_; ++i)
            for (std::size_t q = 0; q < r; ++q) ref(q, i) = desired_ctrl_traj_total(timestep + i, q);
        std::vector<double> u(m);
        mpcx_dims d = control_sys_rs::detail::dims(n, m, r, f_, v_, 1, 1);
        d.u_shared = 1;
        control_sys_rs::detail::Device::check(
            mpcx_closed_loop_f64(control_sys_rs::detail::Device::ctx(), &d, a.as_ptr(), b.as_ptr(), c.as_ptr(), state.as_ptr(),
                                 mat_o_.as_ptr(), gain_matrix_.as_ptr(), ref.as_ptr(), nullptr, u.data(), nullptr),
            "mpcx_closed_loop_f64");
        return u;
    }

    std::shared_ptr<const model::DiscreteStateSpaceModel> model_;
    std::size_t f_ = 0, v_ = 0;
    na::DMatrix mat_o_, gain_matrix_;
};

}  // namespace mpc
}  // namespace controller

// ------------------------------------------------------------------------------------------------ additive batched API
// Many independent plants per call: where the GPU pays.  Arrays hold `batch` matrices back to back, each column-major
// (one DMatrix::as_slice() after the other), i.e. the instance-major layout of the C ABI; the library re-arranges them
// on the device for the kernels that prefer the batch-interleaved layout.
namespace batch {

struct QpBatch {
    std::size_t n = 0, m = 0, r = 0, f = 0, v = 0, batch = 0;
    std::vector<double> phi, gamma, h, fvec;  // (f r) x n, (f r) x (v m), (v m) x (v m), v m per plant
};

// Phi, Gamma, H = Gamma^T blockdiag(Q) Gamma + W3 and fvec = Gamma^T blockdiag(Q) (Phi x0 - ref) for every plant.
// q is the r x r block of W4 (Controller::new: every element = pred_weight), w3 the shared (v m) x (v m) matrix.
inline QpBatch build_qp_batched(std::size_t batch, std::size_t n, std::size_t m, std::size_t r, std::size_t f, std::size_t v,
                                const std::vector<double> &a, const std::vector<double> &b, const std::vector<double> &c,
                                const na::DMatrix &w3, const na::DMatrix &q, const std::vector<double> &x0,
                                const std::vector<double> &reference) {
    if (a.size() != batch * n * n || b.size() != batch * n * m || c.size() != batch * r * n || x0.size() != batch * n ||
        reference.size() != batch * f * r || w3.nrows() != v * m || w3.ncols() != v * m || q.nrows() != r || q.ncols() != r)
        throw Panic("Matrix multiplication dimensions mismatch.");
    QpBatch out;
    out.n = n; out.m = m; out.r = r; out.f = f; out.v = v; out.batch = batch;
    out.phi.resize(batch * f * r * n);
    out.gamma.resize(batch * f * r * v * m);
    out.h.resize(batch * v * m * v * m);
    out.fvec.resize(batch * v * m);
    const mpcx_dims d = detail::dims(n, m, r, f, v, (std::int64_t)batch);
    detail::Device::check(mpcx_qp_build_f64(detail::Device::ctx(), &d, a.data(), b.data(), c.data(), w3.as_ptr(), q.as_ptr(),
                                            x0.data(), reference.data(), out.phi.data(), out.gamma.data(), out.h.data(),
                                            out.fvec.data()), "mpcx_qp_build_f64");
    return out;
}

// X (n x (T+1)) and Y (r x T) per plant; u holds m x T per plant, or one m x T shared by all when `shared_input`.
inline std::pair<std::vector<double>, std::vector<double>> simulate_batched(std::size_t batch, std::size_t n, std::size_t m,
                                                                            std::size_t r, std::size_t T,
                                                                            const std::vector<double> &a, const std::vector<double> &b,
                                                                            const std::vector<double> &c, const std::vector<double> &x0,
                                                                            const std::vector<double> &u, bool shared_input) {
    if (a.size() != batch * n * n || b.size() != batch * n * m || c.size() != batch * r * n || x0.size() != batch * n ||
        u.size() != (shared_input ? 1 : batch) * m * T)
        throw Panic("Matrix multiplication dimensions mismatch.");
    std::vector<double> x(batch * n * (T + 1)), y(batch * r * T);
    mpcx_dims d = detail::dims(n, m, r, 1, 1, (std::int64_t)batch, (std::int64_t)T);
    d.u_shared = shared_input ? 1 : 0;
    detail::Device::check(mpcx_rollout_f64(detail::Device::ctx(), &d, a.data(), b.data(), c.data(), x0.data(), u.data(), x.data(),
                                           y.data()), "mpcx_rollout_f64");
    return {std::move(x), std::move(y)};
}

// gain = H^-1 Gamma^T blockdiag(Q) per plant and the status the reference reports as Err("Matrix inversion failed")
// (info[b] = 1); the batch continues past a singular plant.
inline std::pair<std::vector<double>, std::vector<std::int32_t>> gain_batched(const QpBatch &qp, const na::DMatrix &q) {
    std::vector<double> gain(qp.batch * qp.v * qp.m * qp.f * qp.r);
    std::vector<std::int32_t> info(qp.batch, -1);
    const mpcx_dims d = detail::dims(qp.n, qp.m, qp.r, qp.f, qp.v, (std::int64_t)qp.batch);
    detail::Device::check(mpcx_qp_gain_f64(detail::Device::ctx(), &d, qp.h.data(), qp.gamma.data(), q.as_ptr(), gain.data(),
                                           info.data()), "mpcx_qp_gain_f64");
    return {std::move(gain), std::move(info)};
}

// The first m rows of every plant's gain (m x (f r) per plant): what the control move applies (controller.rs:130-132),
// for the cost of the factorisation alone.  Same status array.
inline std::pair<std::vector<double>, std::vector<std::int32_t>> gain_rows_batched(const QpBatch &qp, const na::DMatrix &q) {
    std::vector<double> rows(qp.batch * qp.m * qp.f * qp.r);
    std::vector<std::int32_t> info(qp.batch, -1);
    const mpcx_dims d = detail::dims(qp.n, qp.m, qp.r, qp.f, qp.v, (std::int64_t)qp.batch);
    detail::Device::check(mpcx_qp_gain_rows_f64(detail::Device::ctx(), &d, qp.h.data(), qp.gamma.data(), q.as_ptr(), rows.data(),
                                                info.data()), "mpcx_qp_gain_rows_f64");
    return {std::move(rows), std::move(info)};
}

}  // namespace batch

}  // namespace control
}  // namespace control_sys_rs

#endif  // CONTROL_SYS_RS_HPP
