// adapter_bench.cpp -- the drop-in path timed end to end: include/rts_adapter.hpp (the rs::RTS replacement) over the mock
// SOARS World, ONE C++ process, no Python: scene set-up, then `pulses` transmitter pulses of W^3 launch indices against an
// icosphere target of 20 * 4^subdivs triangles (6 -> 81 920: the scale of BASELINE configs[2]) that moves every pulse, four
// receivers, the simulator's RCS / gain callbacks on the host for every received ray, aggregation and Response emission
// per pulse -- everything rs::RTS does per CPI (ray_tracer.cpp:806-1336).
//   adapter_bench [W=216] [pulses=64] [in_flight=3] [subdivs=6] [maxRefl=6] [device_build=0]
// prints one JSON line: run times for `pulses` and 4 x `pulses`, the marginal ms per pulse, the set-up per run, Mrays/s.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include "mock_soars.hpp"
#include "rts_adapter.hpp"

int main(int argc, char** argv)
{
    using namespace mock;
    const unsigned W = argc > 1 ? (unsigned)atoi(argv[1]) : 216u, pulses = argc > 2 ? (unsigned)atoi(argv[2]) : 64u;
    const unsigned in_flight = argc > 3 ? (unsigned)atoi(argv[3]) : 3u, subdivs = argc > 4 ? (unsigned)atoi(argv[4]) : 6u, max_refl = argc > 5 ? (unsigned)atoi(argv[5]) : 6u;
    Params::vars = {W, max_refl, 0};
    World w;
    Transmitter tx; tx.pos = Vec3(-2000, 0, 0); tx.span = D3{0.04, 0.04, 0.05}; tx.pulses = pulses;      // the sphere (r = 15 m at 2 km) fills ~14 % of the beam
    Receiver rx[4];
    for (int k = 0; k < 4; k++) {
        const double a = -0.6 + 0.4 * k;
        rx[k].pos = Vec3(-2000.0 * std::cos(a), 2000.0 * std::sin(a), 20.0 * k);
        rx[k].ant.az = std::atan2(-rx[k].pos.y, -rx[k].pos.x); rx[k].ant.el = std::atan2(-rx[k].pos.z, std::hypot(rx[k].pos.x, rx[k].pos.y)); rx[k].ant.gk = 0.5;
        rx[k].sphere = D3{50.0, 2.6, 2.6};
    }
    Target s; s.shape = "sphere"; s.subdivs = subdivs; s.radius = 15.0f; s.p0 = Vec3(0, 0, 0); s.vel = Vec3(200, 20, 0); s.refl = 0.9;
    s.rotating = true; s.rate = YPR{1.0, 0, 0}; s.ra = 0.3; s.rb = 0.1;     // angle-dependent RCS and receive gain: the host callbacks do real work
    w.transmitters = {&tx}; w.receivers = {&rx[0], &rx[1], &rx[2], &rx[3]}; w.targets = {&s};
    RtsStats st{};
    rts_amd::RunOptions opt; opt.in_flight = in_flight; opt.devices.assign(1, 0); opt.last_stats = &st;
    if (argc > 6 && atoi(argv[6])) opt.flags |= RTS_FLAG_DEVICE_BUILD;                                   // [device_build=0]
    // run 0: a few pulses (first launches of a handle: cold caches, tiles in index order); runs 1 and 2: `pulses` and 4 x `pulses`
    // pulses -- every run() sets the scene up again (meshes, hierarchy), as rs::RTS does per call, so the difference of the two
    // gives the marginal cost of a pulse and the rest is the set-up
    auto timed_run = [&](unsigned n) -> double {
        for (auto* r : w.receivers) { for (auto* q : r->responses) delete q; r->responses.clear(); }
        tx.pulses = n;
        const auto a = std::chrono::steady_clock::now();
        rts_amd::run<mock::Traits>(&w, 1024, 65535, opt);
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
    };
    try {
        const double warm = timed_run(6), r1 = timed_run(pulses), r4 = timed_run(4 * pulses);
        size_t n_resp = 0;
        for (auto* r : w.receivers) n_resp += r->responses.size();
        const double per_pulse = (r4 - r1) / (3.0 * pulses), setup = r1 - per_pulse * pulses;
        printf("{\"what\": \"rts_amd::run<Traits> (C++ adapter, mock SOARS world)\", \"W\": %u, \"pulses\": [%u, %u], \"in_flight\": %u, \"triangles\": %u, \"max_refl\": %u, "
               "\"first_run_6_pulses_s\": %.3f, \"run_s\": [%.4f, %.4f], \"marginal_ms_per_pulse\": %.4f, \"setup_s_per_run\": %.4f, \"segments_last_pulse\": %llu, "
               "\"received_last_pulse\": %llu, \"marginal_Mrays_per_s\": %.1f, \"responses_last_run\": %zu}\n",
               W, pulses, 4 * pulses, in_flight, 20u << (2 * subdivs), max_refl, warm, r1, r4, per_pulse * 1e3, setup, (unsigned long long)st.segments,
               (unsigned long long)st.received, (double)st.segments / per_pulse / 1e6, n_resp);
    } catch (const std::exception& e) { fprintf(stderr, "adapter failed: %s\n", e.what()); return 2; }
    return 0;
}
