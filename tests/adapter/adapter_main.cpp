// adapter_main.cpp -- drives include/rts_adapter.hpp (the rs::RTS replacement) over the mock World described by a scenario
// file and prints every emitted response, one per line:
//     R tx-index pulse-time rx power delay doppler phase noise
// then the receivers' noise temperatures after the run (SetNoiseTemperature accumulates once per TRANSMITTER,
// ray_tracer.cpp:829):
//     N rx noise
// Usage: adapter_main SCENARIO [in_flight=2] [handle_sets=1] [rays]
//   in_flight    pulses in flight per handle set (1 = sequential)
//   handle_sets  number of handle SETS, all on device 0 -- the multi-device path of the adapter on one GPU
//   rays         every pulse split over all handle sets (interleaved tiles); default: whole pulses per set
// Scenario file (written by tests/adapter_ref.py from the same dict its oracle-side driver reads): one object per line,
//     params|tx|rx|target key=value ...
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include "mock_soars.hpp"
#include "rts_adapter.hpp"

namespace {
struct KV {
    std::map<std::string, std::string> m;
    double d(const char* k, double def = 0) const { auto it = m.find(k); return it == m.end() ? def : strtod(it->second.c_str(), nullptr); }
    unsigned u(const char* k, unsigned def = 0) const { auto it = m.find(k); return it == m.end() ? def : (unsigned)strtoul(it->second.c_str(), nullptr, 10); }
    std::string s(const char* k, const char* def = "") const { auto it = m.find(k); return it == m.end() ? std::string(def) : it->second; }
};
void read_antenna(const KV& kv, mock::Antenna& a) {
    a.az = kv.d("az"); a.el = kv.d("el"); a.az_rate = kv.d("az_rate"); a.el_rate = kv.d("el_rate"); a.wob = kv.d("wob"); a.wob_w = kv.d("wob_w");
    a.g0 = kv.d("g0", 1); a.gk = kv.d("gk"); a.gw = kv.d("gw");
}
}

int main(int argc, char** argv)
{
    using namespace mock;
    if (argc < 2) { fprintf(stderr, "usage: adapter_main SCENARIO [in_flight] [handle_sets] [rays]\n"); return 1; }
    std::ifstream in(argv[1]);
    if (!in) { fprintf(stderr, "cannot open %s\n", argv[1]); return 1; }
    World w;
    std::vector<std::unique_ptr<Transmitter>> txs; std::vector<std::unique_ptr<Receiver>> rxs; std::vector<std::unique_ptr<Target>> tgs;
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream ls(line); std::string kind, tok; KV kv;
        if (!(ls >> kind) || kind[0] == '#') continue;
        while (ls >> tok) { const size_t e = tok.find('='); if (e != std::string::npos) kv.m[tok.substr(0, e)] = tok.substr(e + 1); }
        if (kind == "params") {
            Params::vars = {kv.u("W", 16), kv.u("max_refl", 4), kv.u("max_refr", 0)};
            Params::c_ = kv.d("c", 299792458.0); Params::start_ = kv.d("start"); Params::rate_ = kv.d("rate", 1000.0); Params::smooth_ = kv.u("smooth", 1) != 0;
        } else if (kind == "tx") {
            auto t = std::make_unique<Transmitter>();
            t->pos = Vec3(kv.d("x"), kv.d("y"), kv.d("z")); read_antenna(kv, t->ant);
            t->span = D3{kv.d("span_az"), kv.d("span_el"), kv.d("span_r")}; t->pulses = kv.u("pulses", 1); t->pri = kv.d("pri", 1e-3); t->t_first = kv.d("t_first");
            t->sig.carrier = kv.d("carrier", 10e9); t->sig.temp = kv.d("temp");
            w.transmitters.push_back(t.get()); txs.push_back(std::move(t));
        } else if (kind == "rx") {
            auto r = std::make_unique<Receiver>();
            r->pos = Vec3(kv.d("x"), kv.d("y"), kv.d("z")); read_antenna(kv, r->ant);
            r->sphere = D3{kv.d("radius"), kv.d("span_theta"), kv.d("span_phi")}; r->noise = kv.d("noise", 290);
            w.receivers.push_back(r.get()); rxs.push_back(std::move(r));
        } else if (kind == "target") {
            auto t = std::make_unique<Target>();
            t->shape = kv.s("shape", "sphere"); t->p0 = Vec3(kv.d("x"), kv.d("y"), kv.d("z")); t->vel = Vec3(kv.d("vx"), kv.d("vy"), kv.d("vz"));
            t->rot0 = YPR{kv.d("yaw"), kv.d("pitch"), kv.d("roll")}; t->rate = YPR{kv.d("yaw_rate"), kv.d("pitch_rate"), kv.d("roll_rate")}; t->rotating = kv.u("rotating") != 0;
            t->w = (float)kv.d("w", 1); t->h = (float)kv.d("h", 1); t->d = (float)kv.d("d", 1); t->radius = (float)kv.d("radius", 1); t->subdivs = kv.u("subdivs", 2);
            t->vfile = kv.s("vfile"); t->nfile = kv.s("nfile"); t->refl = kv.d("refl", 0.9); t->refr = kv.d("refr", 1.0);
            t->rcs = kv.d("rcs", 1); t->ra = kv.d("ra"); t->rb = kv.d("rb"); t->rw = kv.d("rw");
            w.targets.push_back(t.get()); tgs.push_back(std::move(t));
        } else { fprintf(stderr, "scenario: unknown object '%s'\n", kind.c_str()); return 1; }
    }
    RtsStats st{};
    rts_amd::RunOptions opt;
    opt.in_flight = argc > 2 ? (unsigned)atoi(argv[2]) : 2u;
    opt.devices.assign(argc > 3 ? (size_t)std::max(1, atoi(argv[3])) : 1u, 0);
    opt.shard_rays = argc > 4 && (std::string(argv[4]) == "rays" || std::string(argv[4]) == "deal");
    opt.deal_after = argc > 4 && std::string(argv[4]) == "deal" ? 2u : 0u;      // from the third pulse on: tile lists dealt from the cost records
    opt.last_stats = &st;
    try { rts_amd::run<mock::Traits>(&w, 1024, 65535, opt); }
    catch (const std::exception& e) { fprintf(stderr, "adapter failed: %s\n", e.what()); return 2; }
    for (size_t j = 0; j < w.receivers.size(); j++)
        for (auto* resp : w.receivers[j]->responses) {
            size_t txi = 0; while (txi < w.transmitters.size() && w.transmitters[txi] != resp->tx) txi++;
            for (auto& pt : resp->pts)
                printf("R %zu %.17g %zu %.17g %.17g %.17g %.17g %.17g\n", txi, pt.time - pt.delay, j, pt.power, pt.delay, pt.doppler, pt.phase, pt.noise);
        }
    for (size_t j = 0; j < w.receivers.size(); j++) printf("N %zu %.17g\n", j, w.receivers[j]->GetNoiseTemperature());
    fprintf(stderr, "last pulse: %llu rays %llu segments %llu received\n", (unsigned long long)st.rays, (unsigned long long)st.segments, (unsigned long long)st.received);
    return 0;
}
