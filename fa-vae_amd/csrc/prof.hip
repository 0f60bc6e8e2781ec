// Launch profiler of libfavae_hip: when enabled, every kernel launch of the library (FAVAE_KLAUNCH, common.h) is bracketed by two
// HIP events recorded on the stream the kernel is launched on, keyed by the kernel's full instantiation name.  bench.py reads
// the per-kernel launch counts / durations / algorithmic work back through favae_prof_report() -- the `roofline` object of its
// JSON line is computed from these, inside the timed region, on the stream each kernel really ran on (the weight gradients run
// on a second stream, which torch.cuda.Event on the current stream would never see).
// Off by default (level 0): one predictable branch per launch.
#include <mutex>
#include <string>
#include <vector>
#include <map>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <cxxabi.h>

#include "common.h"

int favae_prof_level_ = 0;

namespace {
struct Rec {
    const void* fn;                 // host-side kernel function: its device symbol is the name rocprofv3 reports
    hipEvent_t a, b;
    double flops, bytes;
};
std::mutex g_mu;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
thread_local double t_flops = 0.0, t_bytes = 0.0;
thread_local bool t_note = false;

hipEvent_t get_event() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

// device symbol of the kernel, demangled and cut down to "name<template args>" -- the form tools/rocpd_stats.py gives the kernel
// names of a rocprofv3 trace: "void (anonymous namespace)::conv_wgrad_row3_sp_kernel<2, 2>((anonymous namespace)::WgradArgs)"
// -> "conv_wgrad_row3_sp_kernel<2, 2>"
std::string short_name(const void* fn) {
    const char* mangled = hipKernelNameRefByPtr(fn, nullptr);
    if (!mangled) return "?";
    int st = 0;
    char* dem = abi::__cxa_demangle(mangled, nullptr, nullptr, &st);
    std::string s(st == 0 && dem ? dem : mangled);
    free(dem);
    const char* anon = "(anonymous namespace)::";
    for (size_t q; (q = s.find(anon)) != std::string::npos;) s.erase(q, strlen(anon));
    if (s.compare(0, 5, "void ") == 0) s = s.substr(5);
    int depth = 0;
    for (size_t i = 0; i < s.size(); ++i) {
        if (s[i] == '<') ++depth;
        else if (s[i] == '>') --depth;
        else if (s[i] == '(' && depth == 0) { s.resize(i); break; }
    }
    return s;
}
}  // namespace

void favae_prof_note_(double flops, double bytes) {
    t_flops = flops;
    t_bytes = bytes;
    t_note = true;
}

int favae_prof_fail_(int code) {
    t_note = false;
    return code;
}

void* favae_prof_begin_(const void* host_fn, hipStream_t s) {
    const bool noted = t_note;
    const double fl = noted ? t_flops : 0.0, by = noted ? t_bytes : 0.0;
    t_note = false;
    if (favae_prof_level_ < 2 && !(noted && fl >= 1e9)) return nullptr;   // level 1: the matrix-bound launches (>= 1 GFLOP) only
    std::lock_guard<std::mutex> lk(g_mu);
    Rec r{host_fn, get_event(), get_event(), fl, by};
    if (!r.a || !r.b) return nullptr;
    (void)hipEventRecord(r.a, s);
    g_recs.push_back(r);
    return (void*)(uintptr_t)g_recs.size();                           // index + 1
}

void favae_prof_end_(void* rec, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    const size_t i = (size_t)(uintptr_t)rec - 1;
    if (i < g_recs.size()) (void)hipEventRecord(g_recs[i].b, s);
}

extern "C" int favae_prof_enable(int level) {
    if (level < 0 || level > 2) return FAVAE_ERR_BAD_ARG;
    favae_prof_level_ = level;
    t_note = false;                         // a note left by a call that never launched must not be attributed to a later kernel
    return FAVAE_OK;
}

extern "C" int favae_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& r : g_recs) {
        (void)hipEventSynchronize(r.b);
        g_pool.push_back(r.a);
        g_pool.push_back(r.b);
    }
    g_recs.clear();
    return FAVAE_OK;
}

extern "C" int64_t favae_prof_report(char* buf, int64_t cap) {
    std::lock_guard<std::mutex> lk(g_mu);
    struct Agg { long n = 0; double us = 0, mn = 1e30, mx = 0, flops = 0, bytes = 0; };
    std::map<std::string, Agg> agg;
    std::map<const void*, std::string> names;
    for (auto& r : g_recs) {
        if (hipEventSynchronize(r.b) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
        auto it = names.find(r.fn);
        if (it == names.end()) it = names.emplace(r.fn, short_name(r.fn)).first;
        Agg& a = agg[it->second];
        const double us = 1e3 * ms;
        a.n += 1; a.us += us; a.flops += r.flops; a.bytes += r.bytes;
        if (us < a.mn) a.mn = us;
        if (us > a.mx) a.mx = us;
    }
    std::string out;
    char line[512];
    for (auto& kv : agg) {
        snprintf(line, sizeof line, "%s\t%ld\t%.3f\t%.3f\t%.3f\t%.6e\t%.6e\n", kv.first.c_str(), kv.second.n, kv.second.us,
                 kv.second.mn, kv.second.mx, kv.second.flops, kv.second.bytes);
        out += line;
    }
    if (buf && cap > 0) {
        const size_t n = out.size() < (size_t)cap - 1 ? out.size() : (size_t)cap - 1;
        memcpy(buf, out.data(), n);
        buf[n] = 0;
    }
    return (int64_t)out.size() + 1;
}
