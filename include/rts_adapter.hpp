// rts_adapter.hpp -- header-only host driver that gives SOARS/FERS the entry point of the
// reference, rs::RTS(World*, MaxThreads, MaxBlocks) (ray_tracer.cpp:512), on top of the C-ABI
// of librts_amd.so.  It reproduces the CONTROL FLOW of ray_tracer.cpp:806-1336 -- transmitter
// loop, pulse loop, per-pulse target placement, launch, received-ray finalisation with the
// simulator's own antenna-gain and RCS callbacks, aggregation, one Response per unique path --
// and none of its OptiX plumbing.
//
// Use inside SOARS:
//     #include "rts_adapter.hpp"
//     namespace rs { void RTS(World* w, unsigned mt, unsigned mb) { rts_amd::run<SoarsTraits>(w, mt, mb); } }
// where SoarsTraits names the simulator's types (a ready-made rts_amd::SoarsTraits is at the
// bottom, compiled only when RTS_ADAPTER_WITH_SOARS is defined, i.e. with the rs*.cuh headers
// on the include path).  tests/adapter/ instantiates the same template over a mock World.
//
// Differences from the reference that a caller can observe:
//   * errors throw std::runtime_error(rts_last_error()) instead of exit(1);
//   * meshes are built once per target and placed on the device per pulse (the reference
//     rebuilds and re-uploads them every pulse, ray_tracer.cpp:963-1117);
//   * the four wall-clock printf timers are replaced by RtsStats (rts_get_stats);
//   * pulses (or, RunOptions::shard_rays, the rays of every pulse) are spread over all visible GPUs; responses and their
//     order are those of the sequential single-GPU loop.
#ifndef RTS_ADAPTER_HPP
#define RTS_ADAPTER_HPP

#include <algorithm>
#include <chrono>
#include <stdexcept>
#include <string>
#include <vector>
#include "rts_amd.h"

namespace rs {
// exported by librts_amd.so with the reference's exact signature (aggregation.cuh:19-22)
void kernel_wrapper(PerRayData* h_rx_results_arr, int* h_rx_intersects_arr, unsigned int receivedRays,
                    unsigned int depthTotal, unsigned int MaxThreads, unsigned int MaxBlocks, double cspeed,
                    double carrier, double* h_npath_arr, double* h_power_arr, double* h_doppler_arr,
                    double* h_delay_arr, double* h_phase_arr, int* h_pathMatch);
}

namespace rts_amd {

inline void check(int rc, const char* what) {
    if (rc != RTS_OK) throw std::runtime_error(std::string(what) + ": " + rts_last_error());
}

struct HostMesh { std::vector<double> verts, normals; std::vector<uint32_t> tris; };

// Mesh of one target in its own frame, rotated by its t = 0 attitude (ray_tracer.cpp:955-987).
template <class Target>
HostMesh build_target_mesh(Target* targ) {
    HostMesh m;
    const auto rot0 = targ->GetTargetRotation(0);
    const float yaw = (float)rot0.yaw, pitch = (float)rot0.pitch, roll = (float)rot0.roll;   // float, as the reference
    const std::string shape = targ->GetShape();
    if (shape == "rect") {
        float w, h, d; targ->GetRect(w, h, d);
        m.verts.resize(24); m.tris.resize(36); m.normals.resize(36);
        check(rts_rect_mesh(w, h, d, yaw, pitch, roll, m.verts.data(), m.tris.data(), m.normals.data()), "rts_rect_mesh");
    } else if (shape == "sphere") {
        unsigned int subdivs; float radius; targ->GetSphere(subdivs, radius);
        uint32_t nv = 0, nt = 0;
        check(rts_sphere_mesh(subdivs, radius, yaw, pitch, roll, nullptr, &nv, nullptr, &nt, nullptr), "rts_sphere_mesh");
        m.verts.resize(3 * (size_t)nv); m.normals.resize(3 * (size_t)nv); m.tris.resize(3 * (size_t)nt);
        check(rts_sphere_mesh(subdivs, radius, yaw, pitch, roll, m.verts.data(), &nv, m.tris.data(), &nt, m.normals.data()), "rts_sphere_mesh");
    } else if (shape == "file") {
        std::string v_file, n_file; targ->GetFile(v_file, n_file);
        uint32_t nt = 0;
        check(rts_file_mesh(v_file.c_str(), n_file.c_str(), yaw, pitch, roll, nullptr, nullptr, nullptr, &nt), "rts_file_mesh");
        m.verts.resize(9 * (size_t)nt); m.normals.resize(9 * (size_t)nt); m.tris.resize(3 * (size_t)nt);
        check(rts_file_mesh(v_file.c_str(), n_file.c_str(), yaw, pitch, roll, m.verts.data(), m.tris.data(), m.normals.data(), &nt), "rts_file_mesh");
    } else {
        throw std::runtime_error("rts_adapter: unknown target shape '" + shape + "'");
    }
    return m;
}

// Optional instrumentation of a run (RunOptions::times): when each pulse was FINISHED (its responses emitted), on the steady
// clock, in seconds since run() was entered, in pulse order over all transmitters -- the differences are the pulse rate the
// simulator sees, set-up excluded -- and where the submitting thread spent its time, per section of the pulse loop.
struct RunTimes {
    std::vector<double> pulse_done_s;
    double setup_s = 0;                    // run() entry -> the first pulse is begun (handles, meshes, hierarchy)
    enum { BEGIN_HOST = 0, BEGIN_CALL, END_WAIT, GET_RECEIVED, CALLBACKS, AGGREGATE, RESPONSES, N_LAPS };
    double lap_s[N_LAPS] = {0, 0, 0, 0, 0, 0, 0};      // simulator calls + rts_set_receivers | rts_trace_pulse_begin | rts_trace_pulse_end (wait for the trace) |
                                           // rts_get_received | RCS / gain callbacks | aggregation round trip | unique paths + Response emission
    static const char* lap_name(int k) { static const char* n[] = {"begin_host", "trace_pulse_begin", "end_wait", "get_received", "callbacks", "aggregate", "responses"}; return n[k]; }
};

// How the pulses of a run are spread over the GPUs of the machine (the reference is single-GPU).
struct RunOptions {
    std::vector<int> devices;      // HIP device ordinals, one SET of handles each; empty = every visible device.  An ordinal may
                                   // repeat: two handle sets on one GPU (that is how the multi-device path is tested on one GPU)
    unsigned in_flight = 2;        // pulses in flight per handle set (1: no software pipelining)
    bool shard_rays = false;       // false: whole pulses are dealt to the handle sets in turn (pulses are independent,
                                   //        ray_tracer.cpp:843); true: EVERY pulse is split over all handle sets in interleaved
                                   //        tiles (rays are independent, ray_tracer.cu:227-253) and its received rays merged
                                   //        on the host in launch-index order -- the way to use N GPUs for ONE pulse
    unsigned deal_after = 0;       // shard_rays: after this many pulses the handle sets' tile cost records are merged and every later pulse is split into
                                   // tile lists DEALT longest-first from them (rts_tile_records_get / rts_deal_tiles / rts_set_tile_list) instead of the
                                   // static interleave; 0 = never.  Results do not depend on it (any partition of the launch indices gives the same rows)
    unsigned flags = 0;            // RtsParams.flags of every handle: RTS_FLAG_DEVICE_BUILD builds the hierarchy on the GPU in milliseconds
                                   // (a CPI of a few hundred pulses is over before the host SAH build of a large scene has paid for itself)
    RtsStats* last_stats = nullptr;
    RunTimes* times = nullptr;     // optional instrumentation (per-pulse completion stamps, host lap timers)
};

// Traits: the simulator types the driver touches.
//   World, Transmitter, Receiver, Target, TransmitterPulse, RadarSignal, Response, InterpPoint,
//   Vec3 (x,y,z), SVec3 (constructible from Vec3; .length, .azimuth, .elevation), Params (static
//   GetRTSVariables/c/start_time/cw_sample_rate/interpolate_smooth).
template <class Tr>
void run(typename Tr::World* world, unsigned int MaxThreads, unsigned int MaxBlocks, const RunOptions& opt)
{
    using Vec3 = typename Tr::Vec3; using SVec3 = typename Tr::SVec3;
    using Clock = std::chrono::steady_clock;
    const Clock::time_point t_run0 = Clock::now();
    RunTimes* const tm = opt.times;
    auto now_s = [&]() { return std::chrono::duration<double>(Clock::now() - t_run0).count(); };
    // lap(k, t0): adds the time since t0 to section k and returns the new t0 (no-ops without RunOptions::times)
    auto lap = [&](int k, double t0) -> double { if (!tm) return 0.0; const double t1 = now_s(); tm->lap_s[k] += t1 - t0; return t1; };
    if (tm) { tm->pulse_done_s.clear(); tm->setup_s = 0; for (double& v : tm->lap_s) v = 0; }
    bool first_begin = true;
    const auto rts_vars = Tr::Params::GetRTSVariables();                       // ray_tracer.cpp:600-605
    RtsParams params{};
    params.width = rts_vars.x; params.max_refl = rts_vars.y; params.max_refr = rts_vars.z > 0 ? 2u : 0u;
    params.interpolate_smooth = Tr::Params::interpolate_smooth() ? 1u : 0u; params.flags = opt.flags;
    const unsigned D = params.max_refr + params.max_refl;
    const uint64_t launchTotal = (uint64_t)params.width * params.width * params.width;
    const uint64_t rayTotal = launchTotal * (params.max_refr ? params.max_refl + 3 : 1);
    const double cspeed = Tr::Params::c(), sim_starttime = Tr::Params::start_time(), sample_time = 1.0 / Tr::Params::cw_sample_rate();

    auto& transmitters = world->transmitters; auto& receivers = world->receivers; auto& targets = world->targets;
    const uint32_t rxsize = (uint32_t)receivers.size(), targsize = (uint32_t)targets.size();

    // ---- handle sets: S sets (one per entry of opt.devices) x F slots (pulses in flight per set).  Slot 0 of a set owns the
    // set's scene, the other slots share it (rts_share_scene: one hierarchy per set, built once).
    std::vector<int> devices = opt.devices;
    if (devices.empty()) { int n = 0; check(rts_device_count(&n), "rts_device_count"); for (int i = 0; i < n; i++) devices.push_back(i); }
    if (devices.empty()) throw std::runtime_error("rts_adapter: no HIP device");
    const unsigned S = (unsigned)devices.size(), F = opt.in_flight > 1 ? opt.in_flight : 1u;
    // Whole pulses are finished in TWO stages (below): a pulse keeps its handle one pulse longer than it is "in flight", so a set
    // has F + 1 slots -- unless pulses are strictly sequential (F = 1 on one set) or split over the sets (finished in one stage).
    const bool split = opt.shard_rays && S > 1;
    const bool staged = !split && !(S == 1 && F == 1);
    const unsigned Fh = staged ? F + 1 : F;
    struct Handles { std::vector<RtsHandle> h; ~Handles() { for (RtsHandle x : h) rts_destroy(x); } } hs;
    hs.h.assign((size_t)S * Fh, nullptr);
    auto H = [&](unsigned set, unsigned slot) -> RtsHandle& { return hs.h[(size_t)slot * S + set]; };
    for (unsigned f = 0; f < Fh; f++) for (unsigned s = 0; s < S; s++) { params.device = devices[s]; check(rts_create(&params, &H(s, f)), "rts_create"); }

    // scene: once (the reference regenerates identical meshes every pulse)
    std::vector<HostMesh> host(targsize); std::vector<RtsMesh> meshes(targsize);
    for (uint32_t t = 0; t < targsize; t++) {
        host[t] = build_target_mesh(targets[t]);
        meshes[t].triangles = host[t].tris.data(); meshes[t].vertices = host[t].verts.data(); meshes[t].normals = host[t].normals.data();
        meshes[t].n_triangles = (uint32_t)(host[t].tris.size() / 3); meshes[t].n_vertices = (uint32_t)(host[t].verts.size() / 3);
        meshes[t].n_normals = (uint32_t)(host[t].normals.size() / 3); meshes[t].reserved = 0;
        meshes[t].refl_coeff = targets[t]->GetReflCoeff(); meshes[t].refr_index = targets[t]->GetRefrIndex();
    }
    for (unsigned s = 0; s < S; s++) {
        check(rts_set_scene(H(s, 0), meshes.data(), targsize), "rts_set_scene");
        for (unsigned f = 1; f < Fh; f++) check(rts_share_scene(H(s, f), H(s, 0)), "rts_share_scene");
    }

    for (size_t tx_i = 0; tx_i < transmitters.size(); tx_i++) {                // ray_tracer.cpp:806
        auto* trans = transmitters[tx_i];
        const unsigned pulseCount = trans->GetPulseCount();
        typename Tr::TransmitterPulse signal_storage; auto* signal = &signal_storage;
        trans->GetPulse(signal, 0);
        auto* wave = signal->wave;
        const double carrier = wave->GetCarrier(), Wl = cspeed / carrier;
        const auto txSpan = trans->GetTxSpan();
        for (uint32_t j = 0; j < rxsize; j++)                                  // side effect kept: once per transmitter (:829)
            receivers[j]->SetNoiseTemperature(wave->GetTemp() + receivers[j]->GetNoiseTemperature());
        const Vec3 trpos = trans->GetPosition(0);                              // Tx position frozen at t = 0 (:881)

        // ---- everything of pulse k up to the launch (:843-1165), left in flight on handle h; item: the part of the pulse's
        // launch indices this handle traces.  Returns the pulse time.
        auto begin_pulse = [&](unsigned k, RtsHandle h, const RtsPlanItem& item) -> double {
            double tl = tm ? now_s() : 0.0;
            if (tm && first_begin) { tm->setup_s = tl; first_begin = false; }
            trans->GetPulse(signal, k);
            const double time_t = signal->time;
            const auto txrot = trans->GetRotation(time_t);
            RtsPulse pulse{};
            pulse.ray_origin[0] = trpos.x; pulse.ray_origin[1] = trpos.y; pulse.ray_origin[2] = trpos.z;
            pulse.tx_span[0] = txSpan.x; pulse.tx_span[1] = txSpan.y; pulse.tx_span[2] = txSpan.z;
            pulse.tx_dir[0] = txrot.azimuth; pulse.tx_dir[1] = txrot.elevation;
            pulse.ray_first = item.ray_first; pulse.ray_count = item.ray_count;
            pulse.interleave_tile = item.interleave_tile; pulse.interleave_parts = item.interleave_parts; pulse.interleave_part = item.interleave_part;

            std::vector<RtsReceiverSphere> spheres(rxsize);                    // :894-918
            for (uint32_t j = 0; j < rxsize; j++) {
                const auto rxrot = receivers[j]->GetRotation(time_t);
                const auto rxsphere = receivers[j]->GetRxSphere();
                const Vec3 repos = receivers[j]->GetPosition(0);
                const double p[3] = {repos.x, repos.y, repos.z};
                check(rts_rx_sphere(p, rxrot.azimuth, rxrot.elevation, rxsphere.x, rxsphere.y, rxsphere.z, &spheres[j]), "rts_rx_sphere");
            }
            check(rts_set_receivers(h, spheres.data(), rxsize), "rts_set_receivers");

            std::vector<RtsTargetMotion> motion(targsize);                     // :936-1014, 1144-1145
            for (uint32_t t = 0; t < targsize; t++) {
                const Vec3 p0 = targets[t]->GetPosition(time_t), p1 = targets[t]->GetPosition(time_t + sample_time);
                RtsTargetMotion& m = motion[t]; m = RtsTargetMotion{};
                m.position[0] = p0.x; m.position[1] = p0.y; m.position[2] = p0.z;
                m.velocity[0] = (p1.x - p0.x) / sample_time; m.velocity[1] = (p1.y - p0.y) / sample_time; m.velocity[2] = (p1.z - p0.z) / sample_time;
                if (targets[t]->GetRotating() && time_t > sim_starttime) {
                    const auto r = targets[t]->GetTargetRotation(time_t);
                    check(rts_rotation_matrix((float)r.yaw, (float)r.pitch, (float)r.roll, m.rotation), "rts_rotation_matrix");
                    m.has_rotation = 1;
                }
            }
            pulse.motion = motion.data();
            tl = lap(RunTimes::BEGIN_HOST, tl);
            check(rts_trace_pulse_begin(h, &pulse), "rts_trace_pulse_begin");   // replaces :1126-1165
            // a whole pulse: its received set comes home through the handle's pinned mirror, enqueued behind the trace
            if (!split) check(rts_received_prefetch(h), "rts_received_prefetch");
            lap(RunTimes::BEGIN_CALL, tl);
            return time_t;
        };

        // ---- read-back, finalisation, aggregation and responses of one pulse whose launch indices were traced by the handles
        // `parts` (one handle: a whole pulse; several: interleaved parts, merged here in launch-index order) (:1180-1321)
        auto finish_pulse = [&](const std::vector<RtsHandle>& parts, double time_t) {
            double tl = tm ? now_s() : 0.0;
            struct Done { RunTimes* t; decltype(now_s)& now; ~Done() { if (t) t->pulse_done_s.push_back(now()); } } done_stamp{tm, now_s};      // (every way out of this pulse)
            uint64_t R = 0; std::vector<uint64_t> Rp(parts.size(), 0);
            for (size_t q = 0; q < parts.size(); q++) {
                check(rts_trace_pulse_end(parts[q]), "rts_trace_pulse_end");
                check(rts_received_count(parts[q], &Rp[q]), "rts_received_count"); R += Rp[q];
            }
            tl = lap(RunTimes::END_WAIT, tl);
            if (opt.last_stats) rts_get_stats(parts[0], opt.last_stats);
            if (R == 0) return;
            if (R > 0x7ffffffeULL) throw std::runtime_error("rts_adapter: more than 2^31 received rays in one pulse");
            std::vector<PerRayData> rx_results(R); std::vector<int> rx_intersects((size_t)R * D); std::vector<double> rcs_angle((size_t)R * D * 2);
            if (parts.size() == 1) {
                check(rts_get_received(parts[0], rx_results.data(), rx_intersects.data(), rcs_angle.data(), nullptr, R), "rts_get_received");
            } else {        // every part's list ascends in launch index (slot): merge the lists into one ascending list
                std::vector<PerRayData> rr(R); std::vector<int> ri((size_t)R * D); std::vector<double> ra((size_t)R * D * 2); std::vector<uint64_t> slots(R);
                uint64_t off = 0;
                for (size_t q = 0; q < parts.size(); q++) {
                    if (Rp[q]) check(rts_get_received(parts[q], rr.data() + off, ri.data() + off * D, ra.data() + off * D * 2, slots.data() + off, Rp[q]), "rts_get_received");
                    off += Rp[q];
                }
                std::vector<uint64_t> order(R); for (uint64_t i = 0; i < R; i++) order[i] = i;
                // rows of one launch index (refraction: chains k W^3 apart) come from the same part; (slot mod W^3, slot) keeps
                // the reference's order, which is the buffer-row order = ascending slot
                std::stable_sort(order.begin(), order.end(), [&](uint64_t x, uint64_t y) { return slots[x] < slots[y]; });
                for (uint64_t i = 0; i < R; i++) {
                    const uint64_t j = order[i];
                    rx_results[i] = rr[j];
                    for (unsigned d = 0; d < D; d++) { rx_intersects[(size_t)i * D + d] = ri[(size_t)j * D + d]; rcs_angle[((size_t)i * D + d) * 2] = ra[((size_t)j * D + d) * 2]; rcs_angle[((size_t)i * D + d) * 2 + 1] = ra[((size_t)j * D + d) * 2 + 1]; }
                }
            }

            tl = lap(RunTimes::GET_RECEIVED, tl);
            const Vec3 origin = trpos;
            for (uint64_t i = 0; i < R; i++) {                                 // :1198-1256 for the received rays
                PerRayData& r = rx_results[i];
                auto* recv = receivers[r.received];
                const Vec3 repos = recv->GetPosition(0);
                SVec3 transvec, recvvec;
                if (r.reflDepth == 0 && r.refrDepth == 0) {
                    transvec = SVec3(Vec3(origin.x - repos.x, origin.y - repos.y, origin.z - repos.z));
                    recvvec = SVec3(Vec3(repos.x - origin.x, repos.y - origin.y, repos.z - origin.z));
                } else {
                    transvec = SVec3(Vec3(r.firstHitPoint.x - origin.x, r.firstHitPoint.y - origin.y, r.firstHitPoint.z - origin.z));
                    recvvec = SVec3(Vec3(r.prevHitPoint.x - repos.x, r.prevHitPoint.y - repos.y, r.prevHitPoint.z - repos.z));
                }
                transvec.length = 1; recvvec.length = 1;
                const double delay = r.rayLength / cspeed;
                for (unsigned d = 0; d < D; d++) {
                    const int targ_k = rx_intersects[(size_t)i * D + d];
                    if (targ_k >= 0) r.power *= targets[targ_k]->GetRCS(rcs_angle[((size_t)i * D + d) * 2], rcs_angle[((size_t)i * D + d) * 2 + 1], Wl);
                }
                const double Gt = trans->GetGain(transvec, trans->GetRotation(time_t), Wl);
                const double Gr = recv->GetGain(recvvec, recv->GetRotation(delay + time_t), Wl);
                r.power *= (Wl * Wl * Gt * Gr);
                const double Vr = r.doppler / 2;
                r.doppler = carrier * (((1 + Vr / cspeed) / (1 - Vr / cspeed)) - 1);
            }

            tl = lap(RunTimes::CALLBACKS, tl);
            std::vector<double> npath(R, 0), power(R, 0), doppler(R, 0), delay(R, 0), phase(R, 0);      // :1266-1271
            std::vector<int> pathMatch(R, (int)std::min<uint64_t>(rayTotal + 1, 0x7fffffffULL));
            // rs::kernel_wrapper's arithmetic (aggregation.cu:103-184) on the device and stream of the handle that traced the
            // pulse; a failure is an exception here (the reference exit(1)s, aggregation.cu:17-27), never a silent skip
            check(rts_kernel_wrapper_on(parts[0], rx_results.data(), rx_intersects.data(), (unsigned)R, D, MaxThreads, MaxBlocks, cspeed, carrier,
                                        npath.data(), power.data(), doppler.data(), delay.data(), phase.data(), pathMatch.data()), "rs::kernel_wrapper");

            tl = lap(RunTimes::AGGREGATE, tl);
            std::vector<int> uniq(pathMatch);                                  // :1290-1292
            std::sort(uniq.begin(), uniq.end()); uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
            for (int i : uniq) {                                               // :1301-1321
                if (i < 0 || (uint64_t)i >= R) throw std::runtime_error("rts_adapter: aggregation returned a representative ray outside the received list");
                const int rx = rx_results[i].received;
                typename Tr::InterpPoint point(rx_results[i].power, time_t + delay[i], delay[i], rx_results[i].doppler, phase[i],
                                               receivers[rx]->GetNoiseTemperature());
                auto* response = new typename Tr::Response(wave, trans);
                response->AddInterpPoint(point);
                receivers[rx]->AddResponse(response);
            }
            lap(RunTimes::RESPONSES, tl);
        };

        // ---- a WHOLE pulse on one handle, finished in two stages that never wait for work they have just enqueued, and without
        // a copy call: stage 1 -- the received set is read where the device left it (rts_received_view: the handle's pinned
        // mirror, stored by a kernel behind the trace), the simulator's callbacks form every ray's power and Doppler shift
        // (:1198-1256), which go back for the aggregation on the device-resident set (rts_finalise_values + rts_aggregate:
        // rs::kernel_wrapper's arithmetic without its eight uploads); stage 2, one pulse later -- the per-ray outputs are read
        // (rts_aggregated_view), unique paths, Responses (:1290-1321).  Same bits as finish_pulse (one device routine serves both).
        struct Stage1 { RtsHandle h; double time_t; uint64_t R; const PerRayData* rays; };
        auto finish_stage1 = [&](RtsHandle h, double time_t) -> Stage1 {
            double tl = tm ? now_s() : 0.0;
            Stage1 o{h, time_t, 0, nullptr};
            const int32_t* rx_intersects = nullptr; const double* rcs_angle = nullptr;
            check(rts_received_view(h, &o.rays, &rx_intersects, &rcs_angle, nullptr, &o.R), "rts_received_view");
            tl = lap(RunTimes::END_WAIT, tl);
            if (opt.last_stats) rts_get_stats(h, opt.last_stats);
            const uint64_t R = o.R;
            if (R == 0) return o;
            if (R > 0x7ffffffeULL) throw std::runtime_error("rts_adapter: more than 2^31 received rays in one pulse");
            tl = lap(RunTimes::GET_RECEIVED, tl);
            std::vector<double> power(R), doppler(R);
            const Vec3 origin = trpos;
            for (uint64_t i = 0; i < R; i++) {                                 // :1198-1256 for the received rays
                const PerRayData& r = o.rays[i];
                auto* recv = receivers[r.received];
                const Vec3 repos = recv->GetPosition(0);
                SVec3 transvec, recvvec;
                if (r.reflDepth == 0 && r.refrDepth == 0) {
                    transvec = SVec3(Vec3(origin.x - repos.x, origin.y - repos.y, origin.z - repos.z));
                    recvvec = SVec3(Vec3(repos.x - origin.x, repos.y - origin.y, repos.z - origin.z));
                } else {
                    transvec = SVec3(Vec3(r.firstHitPoint.x - origin.x, r.firstHitPoint.y - origin.y, r.firstHitPoint.z - origin.z));
                    recvvec = SVec3(Vec3(r.prevHitPoint.x - repos.x, r.prevHitPoint.y - repos.y, r.prevHitPoint.z - repos.z));
                }
                transvec.length = 1; recvvec.length = 1;
                const double delay = r.rayLength / cspeed;
                double pw = r.power;
                for (unsigned d = 0; d < D; d++) {
                    const int targ_k = rx_intersects[(size_t)i * D + d];
                    if (targ_k >= 0) pw *= targets[targ_k]->GetRCS(rcs_angle[((size_t)i * D + d) * 2], rcs_angle[((size_t)i * D + d) * 2 + 1], Wl);
                }
                const double Gt = trans->GetGain(transvec, trans->GetRotation(time_t), Wl);
                const double Gr = recv->GetGain(recvvec, recv->GetRotation(delay + time_t), Wl);
                pw *= (Wl * Wl * Gt * Gr);
                const double Vr = r.doppler / 2;
                power[i] = pw; doppler[i] = carrier * (((1 + Vr / cspeed) / (1 - Vr / cspeed)) - 1);
            }
            tl = lap(RunTimes::CALLBACKS, tl);
            check(rts_finalise_values(h, power.data(), doppler.data(), R), "rts_finalise_values");
            check(rts_aggregate(h, cspeed, carrier, 0), "rts_aggregate");      // rs::kernel_wrapper's arithmetic (aggregation.cu:32-97) on the device-resident set; enqueued
            lap(RunTimes::AGGREGATE, tl);
            return o;
        };
        auto finish_stage2 = [&](const Stage1& o) {
            double tl = tm ? now_s() : 0.0;
            struct Done { RunTimes* t; decltype(now_s)& now; ~Done() { if (t) t->pulse_done_s.push_back(now()); } } done_stamp{tm, now_s};
            if (o.R == 0) return;
            const double *power = nullptr, *doppler = nullptr, *delay = nullptr, *phase = nullptr; const int32_t* pathMatch = nullptr; uint64_t R = 0;
            check(rts_aggregated_view(o.h, &power, &doppler, &delay, &phase, &pathMatch, &R), "rts_aggregated_view");
            if (R != o.R) throw std::runtime_error("rts_adapter: the aggregation returned another number of rays than were received");
            tl = lap(RunTimes::AGGREGATE, tl);
            std::vector<int> uniq(pathMatch, pathMatch + R);                   // :1290-1292
            std::sort(uniq.begin(), uniq.end()); uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
            for (int i : uniq) {                                               // :1301-1321
                if (i < 0 || (uint64_t)i >= R) throw std::runtime_error("rts_adapter: aggregation returned a representative ray outside the received list");
                const int rx = o.rays[i].received;
                typename Tr::InterpPoint point(power[i], o.time_t + delay[i], delay[i], doppler[i], phase[i], receivers[rx]->GetNoiseTemperature());
                auto* response = new typename Tr::Response(wave, trans);
                response->AddInterpPoint(point);
                receivers[rx]->AddResponse(response);
            }
            lap(RunTimes::RESPONSES, tl);
        };

        // ---- the pulse loop (:843), software-pipelined.  Pulses are FINISHED strictly in pulse order, so every side effect
        // (AddResponse) happens in the order of the sequential loop whatever the number of handle sets and slots.
        struct InFlight { std::vector<RtsHandle> parts; double time_t; };
        std::vector<InFlight> fly; size_t done = 0;                            // fly[k] for pulse k; finished up to `done`
        std::vector<Stage1> st1; size_t done2 = 0;                             // staged finish: st1[k] once pulse k's stage 1 has run; stage 2 done up to `done2`
        const unsigned lanes = split ? F : S * F;                              // pulses in flight
        bool dealt = false; uint32_t deal_tile = RTS_PLAN_TILE;
        for (unsigned k = 0; k < pulseCount; k++) {
            InFlight fl;
            if (split && !dealt && opt.deal_after && k == opt.deal_after) {
                // ---- once per run: the sets exchange what every tile cost the set that traced it, and deal the tiles anew (rts_amd.h).
                // No pulse may be in flight on a handle whose list changes: the pipeline is drained first.
                while (done < fly.size()) { finish_pulse(fly[done].parts, fly[done].time_t); done++; }
                const uint32_t n_rec = (uint32_t)((launchTotal + 63) / 64);
                deal_tile = RTS_PLAN_TILE; while (deal_tile > 64u && (launchTotal + deal_tile - 1) / deal_tile < 8ull * S) deal_tile /= 2u;      // (a small lattice: finer tiles, so that every set gets some)
                std::vector<uint32_t> table(n_rec, 0u), mine(n_rec), part_of((size_t)((launchTotal + deal_tile - 1) / deal_tile));
                for (unsigned s = 0; s < S; s++) for (unsigned f = 0; f < F; f++) {     // (sets: disjoint parts; slots of a set: the same part, pulse after pulse)
                    check(rts_tile_records_get(H(s, f), mine.data(), n_rec), "rts_tile_records_get");
                    for (uint32_t i = 0; i < n_rec; i++) if (mine[i] > table[i]) table[i] = mine[i];
                }
                check(rts_deal_tiles(table.data(), n_rec, launchTotal, deal_tile, S, part_of.data(), nullptr), "rts_deal_tiles");
                for (unsigned s = 0; s < S; s++) {
                    std::vector<uint32_t> ids; for (uint32_t t = 0; t < part_of.size(); t++) if (part_of[t] == s) ids.push_back(t);
                    for (unsigned f = 0; f < F; f++) { check(rts_tile_records_set(H(s, f), table.data(), n_rec), "rts_tile_records_set"); check(rts_set_tile_list(H(s, f), deal_tile, ids.data(), (uint32_t)ids.size()), "rts_set_tile_list"); }
                }
                dealt = true;
            }
            if (split) {                                                       // every handle set takes its part of pulse k: interleaved tiles, or the tiles dealt to it
                const unsigned f = k % F;
                for (unsigned s = 0; s < S; s++) {
                    RtsPlanItem item{}; uint32_t n_items = 0;
                    check(rts_plan_cpi(launchTotal, 1, s, S, RTS_SHARD_RAYS, 0, 0, &item, 1, &n_items), "rts_plan_cpi");
                    if (dealt) { item.interleave_tile = deal_tile; item.interleave_parts = RTS_INTERLEAVE_LIST; item.interleave_part = 0; }
                    fl.time_t = begin_pulse(k, H(s, f), item); fl.parts.push_back(H(s, f));
                }
            } else {                                                           // whole pulse on the next (set, slot) in turn
                const unsigned w = k % (S * Fh);
                RtsPlanItem item{}; item.ray_first = 0; item.ray_count = launchTotal;
                fl.time_t = begin_pulse(k, H(w % S, w / S), item); fl.parts.push_back(H(w % S, w / S));
            }
            fly.push_back(fl);
            if (fly.size() - done >= lanes) {
                if (split) finish_pulse(fly[done].parts, fly[done].time_t);
                else {
                    st1.push_back(finish_stage1(fly[done].parts[0], fly[done].time_t));
                    // stage 2 of the pulse BEFORE (its handle is the next one a pulse is begun on); sequential pulses: of this one
                    if (!staged) { finish_stage2(st1[done2]); done2++; }
                    else if (st1.size() - done2 >= 2) { finish_stage2(st1[done2]); done2++; }
                }
                done++;
            }
        }
        while (done < fly.size()) {
            if (split) finish_pulse(fly[done].parts, fly[done].time_t);
            else {
                st1.push_back(finish_stage1(fly[done].parts[0], fly[done].time_t));
                while (st1.size() - done2 >= (staged ? 2u : 1u)) { finish_stage2(st1[done2]); done2++; }
            }
            done++;
        }
        while (done2 < st1.size()) { finish_stage2(st1[done2]); done2++; }
    }
}

// single handle set on one device (the signature of round 1)
template <class Tr>
void run(typename Tr::World* world, unsigned int MaxThreads, unsigned int MaxBlocks, int device = 0, RtsStats* last_stats = nullptr, unsigned in_flight = 2)
{
    RunOptions opt; opt.devices = {device}; opt.in_flight = in_flight; opt.last_stats = last_stats;
    run<Tr>(world, MaxThreads, MaxBlocks, opt);
}

}  // namespace rts_amd

#ifdef RTS_ADAPTER_WITH_SOARS
// Inside SOARS (rsworld.cuh, rsradar.cuh, rstarget.cuh, rsparameters.cuh, rsresponse.cuh, rspath.cuh on the include path)
namespace rts_amd {
struct SoarsTraits {
    using World = rs::World; using TransmitterPulse = rs::TransmitterPulse; using Response = rs::Response;
    using InterpPoint = rs::InterpPoint; using Vec3 = rs::Vec3; using SVec3 = rs::SVec3; using Params = rs::rsParameters;
};
}
// every visible GPU, whole pulses dealt to them in turn (RunOptions{}); SOARS' call site is unchanged
namespace rs { inline void RTS(World* world, unsigned int MaxThreads, unsigned int MaxBlocks) { rts_amd::run<rts_amd::SoarsTraits>(world, MaxThreads, MaxBlocks, rts_amd::RunOptions{}); } }
#endif

#endif  // RTS_ADAPTER_HPP
