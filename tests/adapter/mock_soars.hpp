// mock_soars.hpp -- the smallest stand-in for the SOARS/FERS classes that rs::RTS touches
// (named at ray_tracer.cpp:50-60, absent from the reference repository).  Test scaffolding for
// include/rts_adapter.hpp.  Antenna gains depend on the look direction AND on the antenna's rotation at the time
// asked for, rotations change with time (so Gr at `delay + time_t` differs from Gr at `time_t`), the RCS depends on
// both bistatic angles and on the wavelength: a driver that hands any of ray_tracer.cpp:1204-1247's arguments over
// wrongly (first / previous hit point, the row of rcs_angle, the time of the rotation) emits different responses.
// tests/adapter_ref.py restates the same patterns in Python, independently, for the oracle side.
#pragma once
#include <cmath>
#include <string>
#include <vector>

namespace mock {
struct Vec3 { double x = 0, y = 0, z = 0; Vec3() {} Vec3(double a, double b, double c) : x(a), y(b), z(c) {} };
struct SVec3 {
    double length = 0, azimuth = 0, elevation = 0;
    SVec3() {}
    explicit SVec3(const Vec3& v) { length = std::sqrt(v.x*v.x + v.y*v.y + v.z*v.z); azimuth = std::atan2(v.y, v.x); elevation = length > 0 ? std::asin(v.z / length) : 0; }
};
struct D3 { double x, y, z; };
struct U3 { unsigned x, y, z; };
struct YPR { double yaw, pitch, roll; };

struct Params {
    static U3 vars; static double c_, start_, rate_; static bool smooth_;
    static U3 GetRTSVariables() { return vars; }
    static double c() { return c_; }
    static double start_time() { return start_; }
    static double cw_sample_rate() { return rate_; }
    static bool interpolate_smooth() { return smooth_; }
};
inline U3 Params::vars = {16, 4, 0}; inline double Params::c_ = 299792458.0; inline double Params::start_ = 0.0;
inline double Params::rate_ = 1000.0; inline bool Params::smooth_ = true;

struct RadarSignal { double carrier = 10e9, temp = 0; double GetCarrier() const { return carrier; } double GetTemp() const { return temp; } };
struct TransmitterPulse { RadarSignal* wave = nullptr; double time = 0; };

// An antenna: boresight az(t) = az + az_rate t + wob sin(wob_w t), el(t) = el + el_rate t; gain
//   g0 (1 + gw wl) / (1 + gk ((a.az - r.az)^2 + 2 (a.el - r.el)^2))  * a.length      (length is set to 1 by the driver, :1217-1218)
// gk = 0, gw = 0: isotropic g0.
struct Antenna {
    double az = 0, el = 0, az_rate = 0, el_rate = 0, wob = 0, wob_w = 0, g0 = 1, gk = 0, gw = 0;
    SVec3 rotation(double t) const { SVec3 r; r.length = 1; r.azimuth = az + az_rate * t + wob * std::sin(wob_w * t); r.elevation = el + el_rate * t; return r; }
    double gain(const SVec3& a, const SVec3& r, double wl) const {
        const double da = a.azimuth - r.azimuth, de = a.elevation - r.elevation;
        return g0 * (1 + gw * wl) / (1 + gk * (da * da + 2 * de * de)) * a.length;
    }
};

struct Transmitter {
    Vec3 pos; Antenna ant; D3 span{0.1, 0.1, 0.0}; unsigned pulses = 1; double pri = 1e-3, t_first = 0; RadarSignal sig;
    unsigned GetPulseCount() const { return pulses; }
    void GetPulse(TransmitterPulse* p, int k) { p->wave = &sig; p->time = t_first + k * pri; }
    D3 GetTxSpan() const { return span; }
    Vec3 GetPosition(double) const { return pos; }
    SVec3 GetRotation(double t) const { return ant.rotation(t); }
    double GetGain(const SVec3& a, const SVec3& r, double wl) const { return ant.gain(a, r, wl); }
};

struct InterpPoint {
    double power, time, delay, doppler, phase, noise;
    InterpPoint(double p, double t, double d, double f, double ph, double n) : power(p), time(t), delay(d), doppler(f), phase(ph), noise(n) {}
};
struct Response {
    std::vector<InterpPoint> pts; const RadarSignal* wave; const Transmitter* tx;
    Response(const RadarSignal* w, const Transmitter* t) : wave(w), tx(t) {}
    void AddInterpPoint(const InterpPoint& p) { pts.push_back(p); }
};

struct Receiver {
    Vec3 pos; Antenna ant; D3 sphere{50.0, 1.5, 1.5}; double noise = 290; std::vector<Response*> responses;
    double GetNoiseTemperature() const { return noise; }
    void SetNoiseTemperature(double t) { noise = t; }
    D3 GetRxSphere() const { return sphere; }
    Vec3 GetPosition(double) const { return pos; }
    SVec3 GetRotation(double t) const { return ant.rotation(t); }
    double GetGain(const SVec3& a, const SVec3& r, double wl) const { return ant.gain(a, r, wl); }
    void AddResponse(Response* r) { responses.push_back(r); }
    ~Receiver() { for (auto* r : responses) delete r; }
};

// RCS pattern: rcs (1 + ra cos(az) sin(el) + rb cos(2 el) + rw wl)   (ra = rb = rw = 0: constant)
struct Target {
    std::string shape = "sphere"; Vec3 p0, vel; YPR rot0{0, 0, 0}, rate{0, 0, 0}; bool rotating = false;
    float w = 1, h = 1, d = 1, radius = 1; unsigned subdivs = 2; std::string vfile, nfile; double refl = 0.9, refr = 1.0, rcs = 1.0, ra = 0, rb = 0, rw = 0;
    Vec3 GetPosition(double t) const { return Vec3(p0.x + vel.x * t, p0.y + vel.y * t, p0.z + vel.z * t); }
    YPR GetTargetRotation(double t) const { return YPR{rot0.yaw + rate.yaw * t, rot0.pitch + rate.pitch * t, rot0.roll + rate.roll * t}; }
    std::string GetShape() const { return shape; }
    void GetRect(float& a, float& b, float& c) const { a = w; b = h; c = d; }
    void GetSphere(unsigned& s, float& r) const { s = subdivs; r = radius; }
    void GetFile(std::string& v, std::string& n) const { v = vfile; n = nfile; }
    bool GetRotating() const { return rotating; }
    double GetReflCoeff() const { return refl; }
    double GetRefrIndex() const { return refr; }
    double GetRCS(double az, double el, double wl) const { return rcs * (1 + ra * std::cos(az) * std::sin(el) + rb * std::cos(2 * el) + rw * wl); }
};

struct World { std::vector<Transmitter*> transmitters; std::vector<Receiver*> receivers; std::vector<Target*> targets; };

struct Traits {
    using World = mock::World; using TransmitterPulse = mock::TransmitterPulse; using Response = mock::Response;
    using InterpPoint = mock::InterpPoint; using Vec3 = mock::Vec3; using SVec3 = mock::SVec3; using Params = mock::Params;
};
}  // namespace mock
