// MOCK -- a header-only stand-in with the NAMES of the SOARS/FERS classes that rs::RTS touches (listed at
// ray_tracer.cpp:50-60; the real headers are not part of the reference repository).  It exists only so that the
// RTS_ADAPTER_WITH_SOARS branch of include/rts_adapter.hpp (rts_amd::SoarsTraits, rs::RTS) is parsed and type-checked by a
// compiler in this repository's tests; it implements nothing of SOARS.  All six rs*.cuh files of this directory include it.
#pragma once
#include <cmath>
#include <string>
#include <vector>

namespace rs {
struct Vec3 { double x = 0, y = 0, z = 0; Vec3() {} Vec3(double a, double b, double c) : x(a), y(b), z(c) {} };
struct SVec3 { double length = 0, azimuth = 0, elevation = 0; SVec3() {} explicit SVec3(const Vec3& v) { length = std::sqrt(v.x*v.x + v.y*v.y + v.z*v.z); azimuth = std::atan2(v.y, v.x); elevation = length > 0 ? std::asin(v.z / length) : 0; } };
struct stub_uint3 { unsigned x, y, z; };
struct stub_double3 { double x, y, z; };
struct stub_ypr { double yaw, pitch, roll; };
struct rsParameters {
    static stub_uint3 GetRTSVariables() { return stub_uint3{1, 1, 0}; }
    static double c() { return 299792458.0; }
    static double start_time() { return 0; }
    static double cw_sample_rate() { return 1000.0; }
    static bool interpolate_smooth() { return true; }
};
struct RadarSignal { double GetCarrier() const { return 1e10; } double GetTemp() const { return 0; } };
struct TransmitterPulse { RadarSignal* wave = nullptr; double time = 0; };
struct Transmitter {
    unsigned GetPulseCount() const { return 0; }
    void GetPulse(TransmitterPulse*, int) {}
    stub_double3 GetTxSpan() const { return stub_double3{0, 0, 0}; }
    Vec3 GetPosition(double) const { return Vec3(); }
    SVec3 GetRotation(double) const { return SVec3(); }
    double GetGain(const SVec3&, const SVec3&, double) const { return 1; }
};
struct InterpPoint { InterpPoint(double, double, double, double, double, double) {} };
struct Response { Response(const RadarSignal*, const Transmitter*) {} void AddInterpPoint(const InterpPoint&) {} };
struct Receiver {
    double GetNoiseTemperature() const { return 0; } void SetNoiseTemperature(double) {}
    stub_double3 GetRxSphere() const { return stub_double3{1, 1, 1}; }
    Vec3 GetPosition(double) const { return Vec3(); }
    SVec3 GetRotation(double) const { return SVec3(); }
    double GetGain(const SVec3&, const SVec3&, double) const { return 1; }
    void AddResponse(Response*) {}
};
struct Target {
    Vec3 GetPosition(double) const { return Vec3(); }
    stub_ypr GetTargetRotation(double) const { return stub_ypr{0, 0, 0}; }
    std::string GetShape() const { return "sphere"; }
    void GetRect(float&, float&, float&) const {}
    void GetSphere(unsigned&, float&) const {}
    void GetFile(std::string&, std::string&) const {}
    bool GetRotating() const { return false; }
    double GetReflCoeff() const { return 1; } double GetRefrIndex() const { return 1; }
    double GetRCS(double, double, double) const { return 1; }
};
struct World { std::vector<Transmitter*> transmitters; std::vector<Receiver*> receivers; std::vector<Target*> targets; };
}  // namespace rs
