// The reference's demo scenario (src/main.rs:14-71) written against the C++ host API: DC motor, dt = 0.05, step
// response over 10 s, then the 280-step receding-horizon loop with f = v = 20.  The reference draws two charts
// (main.rs:73-188); this program prints the same series as CSV instead: `step,<i>,<y>` and `loop,<t>,<u>,<y>`.
#include <cstdio>
#include <memory>

#include "control_sys_rs.hpp"

using namespace control_sys_rs::control;

int main() {
    try {
        const double sampling_dt = 0.05;
        const model::dc_motor::Parameters params;
        auto plant = std::make_shared<const model::dc_motor::Model>(params, sampling_dt);

        auto [step_response, step, unused_states] = simulator::step(*plant, 10.0);
        (void)step;
        (void)unused_states;

        const std::size_t f = 20, v = 20, time_steps = 300;
        const na::DMatrix trajectory = trajectory::generate_pulse_trajectory(time_steps);
        const na::DVector x0 = na::DVector::zeros(plant->get_mat_a().nrows());

        const double pred_weight = 10.0, q0 = 0.0000000011, q_other = 0.0001;
        auto mpc = controller::mpc::Controller::new_(plant, f, v, q0, q_other, pred_weight, trajectory);

        // main.rs:51-71, one call per step as the reference does it
        std::vector<na::DVector> states{x0};
        std::vector<double> inputs, outputs;
        for (std::size_t t = 0; t < time_steps - f; ++t) {
            const na::DVector input = mpc.compute_control_input(t, states[t]);
            auto [next_state, next_output] = mpc.propagate_dynamics(input, states[t]);
            states.push_back(next_state);
            outputs.push_back(next_output[0]);
            inputs.push_back(input[0]);
        }
        for (std::size_t i = 0; i < step_response.ncols(); ++i) std::printf("step,%zu,%.17g\n", i, step_response(0, i));
        for (std::size_t t = 0; t < inputs.size(); ++t) std::printf("loop,%zu,%.17g,%.17g\n", t, inputs[t], outputs[t]);
        return 0;
    } catch (const control_sys_rs::DeviceError &e) {
        std::fprintf(stderr, "device: %s\n", e.what());
        return 3;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
