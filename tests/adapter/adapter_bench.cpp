// adapter_bench.cpp -- the drop-in path timed end to end: include/rts_adapter.hpp (the rs::RTS replacement) over the mock
// SOARS World, ONE C++ process, no Python: `pulses` transmitter pulses of W^3 launch indices against an icosphere target of
// 20 * 4^subdivs triangles (6 -> 81 920: the scale of BASELINE configs[2]; the same scene as `bench.py --config sphere6`)
// that moves and turns every pulse, four receivers, the simulator's RCS / gain callbacks on the host for every received
// ray, aggregation and Response emission per pulse -- everything rs::RTS does per CPI (ray_tracer.cpp:806-1336).
//   adapter_bench [W=216] [pulses=256] [in_flight=3] [subdivs=6] [maxRefl=6] [intervals=5] [builders=dh]
// One process runs `intervals` intervals of `pulses` pulses for EACH builder named in `builders` (d = hierarchy built on the
// device, the default of the library; h = host SAH builder), alternating d, h, d, h, ... so that both see the same state of
// the box.  Every run() sets its scene up again, as rs::RTS does per call -- that is reported separately (setup_s); the pulse
// rate is taken from the per-pulse completion stamps of the run (rts_amd::RunTimes): the time between the completion of
// pulse `skip` (the pipeline is full, every handle has a tile-cost history) and of the last pulse, and the distribution of
// the gaps between consecutive completions.  Prints one JSON line.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include "mock_soars.hpp"
#include "rts_adapter.hpp"

static double pct(std::vector<double> v, double p) { if (v.empty()) return 0; std::sort(v.begin(), v.end()); const double x = p * (v.size() - 1); const size_t i = (size_t)x; return i + 1 < v.size() ? v[i] + (x - i) * (v[i + 1] - v[i]) : v.back(); }

int main(int argc, char** argv)
{
    using namespace mock;
    const unsigned W = argc > 1 ? (unsigned)atoi(argv[1]) : 216u, pulses = argc > 2 ? (unsigned)atoi(argv[2]) : 256u;
    const unsigned in_flight = argc > 3 ? (unsigned)atoi(argv[3]) : 3u, subdivs = argc > 4 ? (unsigned)atoi(argv[4]) : 6u, max_refl = argc > 5 ? (unsigned)atoi(argv[5]) : 6u;
    const unsigned intervals = argc > 6 ? (unsigned)atoi(argv[6]) : 5u;
    const std::string builders = argc > 7 ? argv[7] : "dh";
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
    rts_amd::RunTimes tm;
    rts_amd::RunOptions opt; opt.in_flight = in_flight; opt.devices.assign(1, 0); opt.last_stats = &st; opt.times = &tm;
    { int node = -1; (void)rts_bind_host_to_device(0, &node); }            // the process on the GPU's socket (what a SOARS host would do once: INTEGRATION.md)
    auto clear_responses = [&]() { for (auto* r : w.receivers) { for (auto* q : r->responses) delete q; r->responses.clear(); } };
    struct Interval { char builder; double run_s, setup_s, ms_per_pulse, gap_med, gap_p10, gap_p90, gap_min, gap_max, lap_ms[rts_amd::RunTimes::N_LAPS]; size_t responses; };
    std::vector<Interval> out;
    const unsigned skip = std::min(pulses / 4u, 32u);
    try {
        {   // warm-up: code objects, first allocations, the process-wide aggregation context
            tx.pulses = 8; opt.flags = 0; rts_amd::run<mock::Traits>(&w, 1024, 65535, opt); clear_responses();
        }
        for (unsigned it = 0; it < intervals; it++) for (char b : builders) {
            tx.pulses = pulses; opt.flags = b == 'h' ? RTS_FLAG_HOST_BUILD : 0u;
            const auto a = std::chrono::steady_clock::now();
            rts_amd::run<mock::Traits>(&w, 1024, 65535, opt);
            Interval iv{}; iv.builder = b; iv.run_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count(); iv.setup_s = tm.setup_s;
            const std::vector<double>& d = tm.pulse_done_s;
            if (d.size() != pulses) { fprintf(stderr, "adapter_bench: %zu completion stamps for %u pulses\n", d.size(), pulses); return 3; }
            std::vector<double> gaps; for (size_t k = skip + 1; k < d.size(); k++) gaps.push_back((d[k] - d[k - 1]) * 1e3);
            iv.ms_per_pulse = (d.back() - d[skip]) / (double)(d.size() - 1 - skip) * 1e3;
            iv.gap_med = pct(gaps, 0.5); iv.gap_p10 = pct(gaps, 0.1); iv.gap_p90 = pct(gaps, 0.9); iv.gap_min = pct(gaps, 0.0); iv.gap_max = pct(gaps, 1.0);
            for (int k = 0; k < rts_amd::RunTimes::N_LAPS; k++) iv.lap_ms[k] = tm.lap_s[k] / pulses * 1e3;
            iv.responses = 0; for (auto* r : w.receivers) iv.responses += r->responses.size();
            clear_responses();
            out.push_back(iv);
        }
    } catch (const std::exception& e) { fprintf(stderr, "adapter failed: %s\n", e.what()); return 2; }
    printf("{\"what\": \"rts_amd::run<Traits> (C++ adapter, mock SOARS world): per-pulse completion stamps\", \"W\": %u, \"pulses_per_interval\": %u, \"skip\": %u, \"in_flight\": %u, "
           "\"triangles\": %u, \"max_refl\": %u, \"segments_last_pulse\": %llu, \"received_last_pulse\": %llu, \"builders\": {", W, pulses, skip, in_flight, 20u << (2 * subdivs), max_refl,
           (unsigned long long)st.segments, (unsigned long long)st.received);
    bool first_b = true;
    for (char b : builders) {
        std::vector<double> v; for (const Interval& iv : out) if (iv.builder == b) v.push_back(iv.ms_per_pulse);
        if (v.empty()) continue;
        const double med = pct(v, 0.5), lo = pct(v, 0.0), hi = pct(v, 1.0);
        printf("%s\"%s\": {\"ms_per_pulse_median\": %.4f, \"min\": %.4f, \"max\": %.4f, \"spread_pct\": %.2f, \"Mrays_per_s_median\": %.1f, \"intervals\": [", first_b ? "" : ", ", b == 'h' ? "host_tree" : "device_tree",
               med, lo, hi, (hi - lo) / med * 100.0, (double)st.segments / (med * 1e-3) / 1e6);
        bool first = true;
        for (const Interval& iv : out) if (iv.builder == b) {
            printf("%s{\"ms_per_pulse\": %.4f, \"gap_ms\": {\"median\": %.4f, \"p10\": %.4f, \"p90\": %.4f, \"min\": %.4f, \"max\": %.4f}, \"run_s\": %.4f, \"setup_s\": %.4f, \"responses\": %zu, \"host_lap_ms_per_pulse\": {",
                   first ? "" : ", ", iv.ms_per_pulse, iv.gap_med, iv.gap_p10, iv.gap_p90, iv.gap_min, iv.gap_max, iv.run_s, iv.setup_s, iv.responses);
            for (int k = 0; k < rts_amd::RunTimes::N_LAPS; k++) printf("%s\"%s\": %.4f", k ? ", " : "", rts_amd::RunTimes::lap_name(k), iv.lap_ms[k]);
            printf("}}");
            first = false;
        }
        printf("]}");
        first_b = false;
    }
    printf("}}\n");
    return 0;
}
