// The reference's own unit tests (cartpole.rs:365-434, mountain_car.rs:347-400, lunar_lander.rs:1557-1606),
// restated against the C++ host mirror (include/mgym.hpp) of the reference interface.  Needs a GPU;
// built and run by tests/test_gpu_host_mirror.py.
#include <math.h>
#include <stdio.h>

#include <memory>

#include "../../include/mgym.hpp"
using namespace mgym_host;

#define REQUIRE(cond) do { if (!(cond)) { printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } } while (0)
template <class F> static bool throws_invalid(F f) { try { f(); } catch (const std::invalid_argument&) { return true; } catch (...) {} return false; }

int main() {
    {   // test_cartpole (cartpole.rs:365-390)
        std::unique_ptr<CartPoleV1> env(CartPoleV1::builder().build());
        auto state = env->reset();
        REQUIRE(state.size() == 4);
        StepInfo si = env->step(0);
        REQUIRE(si.state.size() == 4);
        REQUIRE(si.reward == 1.0f);
        REQUIRE(!si.done);
    }
    {   // test_cartpole_invalid_action (#[should_panic], cartpole.rs:392-403): Discrete(2)
        std::unique_ptr<CartPoleV1> env(CartPoleV1::builder().build());
        env->reset();
        REQUIRE(throws_invalid([&] { env->step(2); }));
    }
    {   // reward_is_one_when_not_terminated (cartpole.rs:405-434)
        std::unique_ptr<CartPoleV1> env(CartPoleV1::builder().build());
        env->reset();
        StepInfo si = env->step(1);
        REQUIRE(si.reward == 1.0f && !si.done);
        bool done = false;
        for (int i = 0; i < 50 && !done; ++i) done = env->step(1).done;
        REQUIRE(done);
    }
    {   // test_mountain_car (mountain_car.rs:347-372) + invalid action (:374-385) + reward (:387-400)
        std::unique_ptr<MountainCarV0> env(MountainCarV0::builder().build());
        auto state = env->reset();
        REQUIRE(state.size() == 2);
        REQUIRE(state[0] >= -0.6f && state[0] < -0.4f && state[1] == 0.0f);
        StepInfo si = env->step(0);
        REQUIRE(si.state.size() == 2 && si.reward == -1.0f && !si.done);
        REQUIRE(throws_invalid([&] { env->step(3); }));
        si = env->step(1);
        REQUIRE(si.reward == -1.0f && !si.done && !si.truncated);
    }
    {   // test_lunar_lander_reset / _step / _actions / _with_wind (lunar_lander.rs:1557-1606)
        std::unique_ptr<LunarLanderV3> env(LunarLanderV3::builder().build());
        bool threw = false;
        try { env->step(0); } catch (const std::logic_error&) { threw = true; }   // "You forgot to call reset()" (:920)
        REQUIRE(threw);
        auto state = env->reset();
        REQUIRE(state.size() == 8);
        StepInfo si = env->step(0);
        REQUIRE(si.state.size() == 8 && !si.done);
        for (uint32_t a = 0; a < 4; ++a) { si = env->step(a); REQUIRE(si.state.size() == 8 && isfinite(si.reward)); }
        std::unique_ptr<LunarLanderV3> windy(LunarLanderV3::builder().enable_wind(true).build());
        windy->reset();
        si = windy->step(2);
        REQUIRE(si.state.size() == 8 && isfinite(si.reward));
        REQUIRE(throws_invalid([] { delete LunarLanderV3::builder().gravity(-12.0f).build(); }));  // :292-296
    }
    printf("host mirror: all reference unit tests passed\n");
    return 0;
}
