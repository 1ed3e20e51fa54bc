// adapter_main.cpp -- drives include/rts_adapter.hpp (the rs::RTS replacement) over the mock World and
// prints every emitted response, one per line:  pulse-time rx power delay doppler phase
// The scene mirrors rts_amd.scenes.config_multi(W=16) with moving targets over 3 pulses.
// argv[1] (optional): pulses in flight per handle set (default 2; 1 = sequential)
// argv[2] (optional): number of handle SETS, all on device 0 (default 1) -- the multi-device path of the adapter on one GPU
// argv[3] (optional): "rays" = every pulse split over all handle sets (interleaved tiles), default whole pulses per set
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <string>
#include "mock_soars.hpp"
#include "rts_adapter.hpp"

int main(int argc, char** argv)
{
    using namespace mock;
    Params::vars = {16, 4, 0};
    World w;
    Transmitter tx; tx.pos = Vec3(-200, 0, 0); tx.span = D3{0.16, 0.12, 0.05}; tx.pulses = 3;
    Receiver r0, r1;
    r0.pos = Vec3(-200, 0, 0); r0.az = 0.0; r0.el = 0.0; r0.sphere = D3{90.0, 2.6, 2.6};
    r1.pos = Vec3(-150, 130, 10); r1.az = std::atan2(-130.0, 150.0); r1.el = std::atan2(-10.0, std::hypot(150.0, 130.0)); r1.sphere = D3{90.0, 2.6, 2.6};
    Target s, b, p;
    s.shape = "sphere"; s.subdivs = 2; s.radius = 4.0f; s.p0 = Vec3(0, 0, 0); s.vel = Vec3(10, 0, 0); s.refl = 0.9;
    b.shape = "rect"; b.w = b.h = b.d = 5.0f; b.rot0 = YPR{0.5, 0.2, 0.1}; b.p0 = Vec3(2, 9, 1); b.vel = Vec3(0, -5, 0); b.refl = 0.8;
    b.rotating = true; b.rate = YPR{30.0, 0, 0};
    p.shape = "rect"; p.w = 0.2f; p.h = 14.0f; p.d = 14.0f; p.rot0 = YPR{0.6, 0, 0}; p.p0 = Vec3(9, -7, 0); p.vel = Vec3(0, 0, 3); p.refl = 0.7;
    w.transmitters = {&tx}; w.receivers = {&r0, &r1}; w.targets = {&s, &b, &p};
    RtsStats st{};
    rts_amd::RunOptions opt;
    opt.in_flight = argc > 1 ? (unsigned)atoi(argv[1]) : 2u;
    opt.devices.assign(argc > 2 ? (size_t)std::max(1, atoi(argv[2])) : 1u, 0);
    opt.shard_rays = argc > 3 && std::string(argv[3]) == "rays";
    opt.last_stats = &st;
    try { rts_amd::run<mock::Traits>(&w, 1024, 65535, opt); }
    catch (const std::exception& e) { fprintf(stderr, "adapter failed: %s\n", e.what()); return 2; }
    for (size_t j = 0; j < w.receivers.size(); j++)
        for (auto* resp : w.receivers[j]->responses)
            for (auto& pt : resp->pts)
                printf("%.17g %zu %.17g %.17g %.17g %.17g\n", pt.time - pt.delay, j, pt.power, pt.delay, pt.doppler, pt.phase);
    fprintf(stderr, "last pulse: %llu rays %llu segments %llu received\n", (unsigned long long)st.rays, (unsigned long long)st.segments, (unsigned long long)st.received);
    return 0;
}
