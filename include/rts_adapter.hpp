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
//   * the four wall-clock printf timers are replaced by RtsStats (rts_get_stats).
#ifndef RTS_ADAPTER_HPP
#define RTS_ADAPTER_HPP

#include <algorithm>
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

// Traits: the simulator types the driver touches.
//   World, Transmitter, Receiver, Target, TransmitterPulse, RadarSignal, Response, InterpPoint,
//   Vec3 (x,y,z), SVec3 (constructible from Vec3; .length, .azimuth, .elevation), Params (static
//   GetRTSVariables/c/start_time/cw_sample_rate/interpolate_smooth).
template <class Tr>
void run(typename Tr::World* world, unsigned int MaxThreads, unsigned int MaxBlocks, int device = 0, RtsStats* last_stats = nullptr, unsigned in_flight = 2)
{
    using Vec3 = typename Tr::Vec3; using SVec3 = typename Tr::SVec3;
    const auto rts_vars = Tr::Params::GetRTSVariables();                       // ray_tracer.cpp:600-605
    RtsParams params{};
    params.width = rts_vars.x; params.max_refl = rts_vars.y; params.max_refr = rts_vars.z > 0 ? 2u : 0u;
    params.interpolate_smooth = Tr::Params::interpolate_smooth() ? 1u : 0u; params.device = device; params.flags = 0;
    const unsigned D = params.max_refr + params.max_refl;
    const uint64_t rayTotal = (uint64_t)params.width * params.width * params.width * (params.max_refr ? params.max_refl + 3 : 1);
    const double cspeed = Tr::Params::c(), sim_starttime = Tr::Params::start_time(), sample_time = 1.0 / Tr::Params::cw_sample_rate();

    auto& transmitters = world->transmitters; auto& receivers = world->receivers; auto& targets = world->targets;
    const uint32_t rxsize = (uint32_t)receivers.size(), targsize = (uint32_t)targets.size();

    // Two handles hold the same scene and take the pulses in turn: while the host finishes pulse k (read-back, RCS/gain
    // loop, aggregation, responses) the device already traces pulse k+1.  Results and the order of every side effect
    // (AddResponse) are those of the sequential loop; in_flight = 1 restores it literally.
    const unsigned n_handles = in_flight > 1 ? 2u : 1u;
    struct Handles { RtsHandle h[2] = {nullptr, nullptr}; ~Handles() { rts_destroy(h[0]); rts_destroy(h[1]); } } hs;
    for (unsigned i = 0; i < n_handles; i++) check(rts_create(&params, &hs.h[i]), "rts_create");

    // scene: once (the reference regenerates identical meshes every pulse)
    std::vector<HostMesh> host(targsize); std::vector<RtsMesh> meshes(targsize);
    for (uint32_t t = 0; t < targsize; t++) {
        host[t] = build_target_mesh(targets[t]);
        meshes[t].triangles = host[t].tris.data(); meshes[t].vertices = host[t].verts.data(); meshes[t].normals = host[t].normals.data();
        meshes[t].n_triangles = (uint32_t)(host[t].tris.size() / 3); meshes[t].n_vertices = (uint32_t)(host[t].verts.size() / 3);
        meshes[t].n_normals = (uint32_t)(host[t].normals.size() / 3); meshes[t].reserved = 0;
        meshes[t].refl_coeff = targets[t]->GetReflCoeff(); meshes[t].refr_index = targets[t]->GetRefrIndex();
    }
    for (unsigned i = 0; i < n_handles; i++) check(rts_set_scene(hs.h[i], meshes.data(), targsize), "rts_set_scene");

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

        // ---- everything of pulse k up to the launch (:843-1165), left in flight on handle h; returns the pulse time
        auto begin_pulse = [&](unsigned k, RtsHandle h) -> double {
            trans->GetPulse(signal, k);
            const double time_t = signal->time;
            const auto txrot = trans->GetRotation(time_t);
            RtsPulse pulse{};
            pulse.ray_origin[0] = trpos.x; pulse.ray_origin[1] = trpos.y; pulse.ray_origin[2] = trpos.z;
            pulse.tx_span[0] = txSpan.x; pulse.tx_span[1] = txSpan.y; pulse.tx_span[2] = txSpan.z;
            pulse.tx_dir[0] = txrot.azimuth; pulse.tx_dir[1] = txrot.elevation;

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
            check(rts_trace_pulse_begin(h, &pulse), "rts_trace_pulse_begin");   // replaces :1126-1165
            return time_t;
        };

        // ---- read-back, finalisation, aggregation and responses of the pulse in flight on handle h (:1180-1321)
        auto finish_pulse = [&](RtsHandle h, double time_t) {
            check(rts_trace_pulse_end(h), "rts_trace_pulse_end");
            if (last_stats) rts_get_stats(h, last_stats);

            uint64_t R = 0; check(rts_received_count(h, &R), "rts_received_count");
            if (R == 0) return;
            std::vector<PerRayData> rx_results(R); std::vector<int> rx_intersects((size_t)R * D); std::vector<double> rcs_angle((size_t)R * D * 2);
            check(rts_get_received(h, rx_results.data(), rx_intersects.data(), rcs_angle.data(), nullptr, R), "rts_get_received");

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

            std::vector<double> npath(R, 0), power(R, 0), doppler(R, 0), delay(R, 0), phase(R, 0);      // :1266-1271
            std::vector<int> pathMatch(R, (int)std::min<uint64_t>(rayTotal + 1, 0x7fffffffULL));
            rs::kernel_wrapper(rx_results.data(), rx_intersects.data(), (unsigned)R, D, MaxThreads, MaxBlocks, cspeed, carrier,
                               npath.data(), power.data(), doppler.data(), delay.data(), phase.data(), pathMatch.data());

            std::vector<int> uniq(pathMatch);                                  // :1290-1292
            std::sort(uniq.begin(), uniq.end()); uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
            for (int i : uniq) {                                               // :1301-1321
                const int rx = rx_results[i].received;
                typename Tr::InterpPoint point(rx_results[i].power, time_t + delay[i], delay[i], rx_results[i].doppler, phase[i],
                                               receivers[rx]->GetNoiseTemperature());
                auto* response = new typename Tr::Response(wave, trans);
                response->AddInterpPoint(point);
                receivers[rx]->AddResponse(response);
            }
        };

        double t_of[2] = {0, 0};
        for (unsigned k = 0; k < pulseCount; k++) {                            // :843, software-pipelined by one pulse
            t_of[k % n_handles] = begin_pulse(k, hs.h[k % n_handles]);
            if (n_handles == 1) finish_pulse(hs.h[0], t_of[0]);
            else if (k > 0) finish_pulse(hs.h[(k - 1) % 2], t_of[(k - 1) % 2]);
        }
        if (n_handles == 2 && pulseCount > 0) finish_pulse(hs.h[(pulseCount - 1) % 2], t_of[(pulseCount - 1) % 2]);
    }
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
namespace rs { inline void RTS(World* world, unsigned int MaxThreads, unsigned int MaxBlocks) { rts_amd::run<rts_amd::SoarsTraits>(world, MaxThreads, MaxBlocks); } }
#endif

#endif  // RTS_ADAPTER_HPP
