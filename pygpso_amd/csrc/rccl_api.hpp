// RCCL entry points, resolved at run time (dlopen) the first time a multi-GPU call is made: a
// single-GPU user never pays the load of librccl, and a process that already holds a copy of it
// (PyTorch ships one under the same SONAME) shares that copy instead of loading a second one.
// RCCL is the only library besides the HIP runtime this engine ever calls.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>
#include <string>

namespace gpso {

struct RcclApi {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string load_error;
  bool ok = false;

  // nullptr on success, else the reason (kept for later calls)
  static RcclApi& get() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] { api.load(); });
    return api;
  }

 private:
  void load() {
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
    if (!h) {
      const char* e = dlerror();
      load_error = std::string("cannot load librccl: ") + (e ? e : "unknown error");
      return;
    }
    auto sym = [&](const char* name) -> void* {
      void* p = dlsym(h, name);
      if (!p && load_error.empty()) load_error = std::string("librccl lacks ") + name;
      return p;
    };
    GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(sym("ncclGetUniqueId"));
    CommInitRank = reinterpret_cast<decltype(CommInitRank)>(sym("ncclCommInitRank"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
    CommAbort = reinterpret_cast<decltype(CommAbort)>(sym("ncclCommAbort"));
    Broadcast = reinterpret_cast<decltype(Broadcast)>(sym("ncclBroadcast"));
    AllGather = reinterpret_cast<decltype(AllGather)>(sym("ncclAllGather"));
    AllReduce = reinterpret_cast<decltype(AllReduce)>(sym("ncclAllReduce"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
    ok = load_error.empty();
  }
};

}  // namespace gpso
