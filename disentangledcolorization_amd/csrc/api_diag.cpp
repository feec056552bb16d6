// api_diag.cpp - profiling (per-stage and per-conv-launch hipEvents) and debugging hooks of a context (split out of api.cpp, round 6).
#include "ctx.h"

extern "C" {

int disco_set_debug_checksums(disco_ctx* c, void* d_table, int rows, int cols) {
    if (!c || rows < 0 || cols < 0) { set_error("disco_set_debug_checksums: bad argument"); return DISCO_EINVAL; }
    std::lock_guard<std::mutex> lk(c->mu);
    c->d_dbg = (unsigned long long*)d_table; c->dbg_rows = d_table ? rows : 0; c->dbg_cols = cols; c->dbg_seq = 0;
    return DISCO_OK;
}

int disco_set_debug_dump(disco_ctx* c, void* d_buf, size_t bytes_per_row) {
    if (!c) return DISCO_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    c->d_dump = (char*)d_buf; c->dump_stride = d_buf ? bytes_per_row : 0;
    return DISCO_OK;
}

int disco_set_profiling(disco_ctx* c, int level) {
    if (!c) return DISCO_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    c->profiling = level;
    return DISCO_OK;
}

int disco_profile_conv(disco_ctx* c, int* launches, float* total_ms, double* total_flops) {
    if (!c || !launches || !total_ms || !total_flops) { set_error("null argument"); return DISCO_EINVAL; }
    *launches = 0; *total_ms = 0.f; *total_flops = 0.0;
    for (auto& e : c->conv_prof) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.e0, e.e1) != hipSuccess) continue;
        *launches += 1; *total_ms += ms; *total_flops += e.flops;
    }
    return DISCO_OK;
}

int disco_profile_conv_bytes(disco_ctx* c, double* total_bytes) {
    if (!c || !total_bytes) { set_error("null argument"); return DISCO_EINVAL; }
    *total_bytes = 0.0;
    for (auto& e : c->conv_prof) *total_bytes += e.bytes;
    return DISCO_OK;
}

int disco_profile_conv_entry(disco_ctx* c, int i, const char** key, float* ms, double* flops) {
    if (!c || i < 0 || i >= (int)c->conv_prof.size() || !key || !ms || !flops) { set_error("bad conv profile index"); return DISCO_EINVAL; }
    auto& e = c->conv_prof[i];
    *key = e.key.c_str(); *flops = e.flops; *ms = -1.f;
    hipEventElapsedTime(ms, e.e0, e.e1);
    return DISCO_OK;
}

int disco_profile_count(disco_ctx* c) {
    if (!c || c->prof.size() < 2) return 0;
    c->prof_ms.clear(); c->prof_flops.clear();
    for (size_t i = 1; i < c->prof.size(); ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->prof[i - 1].ev, c->prof[i].ev) != hipSuccess) ms = -1.f;
        c->prof_ms.push_back({c->prof[i].name, ms});
        c->prof_flops.push_back(c->prof[i].flops);
    }
    return (int)c->prof_ms.size();
}

int disco_profile_entry(disco_ctx* c, int i, const char** name, float* ms, double* flops) {
    if (!c || i < 0 || i >= (int)c->prof_ms.size()) { set_error("bad profile index"); return DISCO_EINVAL; }
    *name = c->prof_ms[i].first.c_str(); *ms = c->prof_ms[i].second; *flops = c->prof_flops[i];
    return DISCO_OK;
}

}  // extern "C"
