// mcr_comm.hpp -- the one collective of the path: RCCL over xGMI, called from the library itself.
//
// Models and parameters are independent (src/mcmc_ref/convert.py:140-147 is an independent loop), so the multi-GPU
// path has no data-path exchange: every rank summarises its share and ONE all-gather of fixed-size per-parameter
// records (16 doubles = 128 bytes) puts every summary on every rank.  The gather is latency-bound
// (packaged corpus: 460 records = 59 KB), so nothing here is tuned for link bandwidth.
//
// librccl is resolved with dlopen on first use: the statistics library keeps loading on hosts without RCCL, and a
// missing or failing RCCL surfaces as MCR_ECOMM with ncclGetErrorString's text.  No MPI, no torch: ranks exchange the
// 128-byte ncclUniqueId themselves (the Python side does it through a file keyed on MASTER_ADDR / MASTER_PORT).
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <vector>

#include "mcr_device.hpp"

namespace mcr {
namespace comm {

inline char* why_buf() { static char why[256] = ""; return why; }

struct Api {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    // the non-blocking mode (a deadline on every collective: mcr_api.hip, comm_wait): optional -- a librccl without
    // them runs the blocking calls as before
    decltype(&ncclCommInitRankConfig) CommInitRankConfig = nullptr;
    decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    bool nonblocking() const { return CommInitRankConfig && CommGetAsyncError && CommAbort; }
};

// Resolved once per process.  Returns nullptr (api_why() says why) when RCCL cannot be loaded.
inline Api* api()
{
    static Api a;
    static bool tried = false;
    if (tried) return a.handle ? &a : nullptr;
    tried = true;
    const char* names[] = {getenv("MCR_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        if (!n || !*n) continue;
        a.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (a.handle) break;
        snprintf(why_buf(), 256, "%s", dlerror());
    }
    if (!a.handle) return nullptr;
    bool ok = true;
    auto sym = [&](const char* name) -> void* {
        void* p = dlsym(a.handle, name);
        if (!p) { ok = false; snprintf(why_buf(), 256, "librccl has no symbol %s", name); }
        return p;
    };
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(sym("ncclAllGather"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(sym("ncclAllReduce"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) { dlclose(a.handle); a.handle = nullptr; return nullptr; }
    a.CommInitRankConfig = reinterpret_cast<decltype(a.CommInitRankConfig)>(dlsym(a.handle, "ncclCommInitRankConfig"));
    a.CommGetAsyncError = reinterpret_cast<decltype(a.CommGetAsyncError)>(dlsym(a.handle, "ncclCommGetAsyncError"));
    a.CommAbort = reinterpret_cast<decltype(a.CommAbort)>(dlsym(a.handle, "ncclCommAbort"));
    return &a;
}

inline const char* api_why() { return why_buf(); }

}  // namespace comm
}  // namespace mcr
