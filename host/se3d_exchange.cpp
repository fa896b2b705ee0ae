// se3d_exchange.cpp — see se3d_exchange.hpp.  RCCL and the HIP runtime are bound at run time (dlopen / dlsym): the host layer builds with plain g++.
#include "se3d_exchange.hpp"

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <thread>

namespace se3d {

namespace {
[[noreturn]] void fail(const std::string& what) { throw std::runtime_error("se3d exchange: " + what); }

void* sym(void* lib, const char* name, const char* libname) {
  void* p = dlsym(lib, name);
  if (!p) fail(std::string("symbol ") + name + " not found in " + libname);
  return p;
}

// ---- HIP runtime (hipError_t is an int-sized enum, hipSuccess = 0; hipMemcpyKind: 1 = host to device, 2 = device to host) ----
struct HipFns {
  int (*Malloc)(void**, size_t) = nullptr;
  int (*Free)(void*) = nullptr;
  int (*SetDevice)(int) = nullptr;
  int (*Memset)(void*, int, size_t) = nullptr;
  int (*MemcpyAsync)(void*, const void*, size_t, int, void*) = nullptr;
  int (*StreamSynchronize)(void*) = nullptr;
  int (*DeviceSynchronize)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  HipFns() {
    // libse3d_hip.so (a link-time dependency of this library) has the runtime loaded already; else look for it
    void* lib = dlsym(RTLD_DEFAULT, "hipMalloc") ? RTLD_DEFAULT : nullptr;
    if (!lib)
      for (const char* n : {"libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6", "/opt/rocm/lib/libamdhip64.so"})
        if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!lib) fail("the HIP runtime (libamdhip64.so) is not loadable; there is no CPU fallback");
    Malloc = reinterpret_cast<decltype(Malloc)>(sym(lib, "hipMalloc", "libamdhip64"));
    Free = reinterpret_cast<decltype(Free)>(sym(lib, "hipFree", "libamdhip64"));
    SetDevice = reinterpret_cast<decltype(SetDevice)>(sym(lib, "hipSetDevice", "libamdhip64"));
    Memset = reinterpret_cast<decltype(Memset)>(sym(lib, "hipMemset", "libamdhip64"));
    MemcpyAsync = reinterpret_cast<decltype(MemcpyAsync)>(sym(lib, "hipMemcpyAsync", "libamdhip64"));
    StreamSynchronize = reinterpret_cast<decltype(StreamSynchronize)>(sym(lib, "hipStreamSynchronize", "libamdhip64"));
    DeviceSynchronize = reinterpret_cast<decltype(DeviceSynchronize)>(sym(lib, "hipDeviceSynchronize", "libamdhip64"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym(lib, "hipGetErrorString", "libamdhip64"));
  }
  void chk(int e, const char* what) const {
    if (e != 0) fail(std::string(what) + ": " + (GetErrorString ? GetErrorString(e) : "HIP error") + " (" + std::to_string(e) + ")");
  }
};
const HipFns& hip() { static HipFns f; return f; }

bool file_exists(const std::string& p) { struct stat st; return ::stat(p.c_str(), &st) == 0; }
void write_atomically(const std::string& path, const void* data, size_t bytes) {
  const std::string tmp = path + ".tmp";
  {
    std::ofstream o(tmp, std::ios::binary | std::ios::trunc);
    o.write(static_cast<const char*>(data), (std::streamsize)bytes);
    if (!o) fail("cannot write " + tmp);
  }
  if (std::rename(tmp.c_str(), path.c_str()) != 0) fail("cannot rename " + tmp);
}
void read_when_there(const std::string& path, void* data, size_t bytes, double timeout_s) {
  const auto t0 = std::chrono::steady_clock::now();
  while (!file_exists(path)) {
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) fail("timed out waiting for " + path + " (a rank died or never started)");
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
  std::ifstream i(path, std::ios::binary);
  i.read(static_cast<char*>(data), (std::streamsize)bytes);
  if (!i || (size_t)i.gcount() != bytes) fail("short read of " + path);
}
}  // namespace

const HipRuntime& HipRuntime::get() { static HipRuntime r; (void)hip(); return r; }
void* HipRuntime::malloc_device(size_t bytes) const { void* p = nullptr; hip().chk(hip().Malloc(&p, bytes ? bytes : 1), "hipMalloc"); return p; }
void HipRuntime::free_device(void* p) const { if (p) (void)hip().Free(p); }
void HipRuntime::set_device(int device) const { hip().chk(hip().SetDevice(device), "hipSetDevice"); }
void HipRuntime::memset_device(void* p, int value, size_t bytes) const { hip().chk(hip().Memset(p, value, bytes), "hipMemset"); hip().chk(hip().DeviceSynchronize(), "hipDeviceSynchronize"); }
void HipRuntime::copy_to_host(void* dst, const void* d_src, size_t bytes, void* stream) const { hip().chk(hip().MemcpyAsync(dst, d_src, bytes, 2, stream), "hipMemcpyAsync (device to host)"); }
void HipRuntime::copy_to_device(void* d_dst, const void* src, size_t bytes, void* stream) const { hip().chk(hip().MemcpyAsync(d_dst, src, bytes, 1, stream), "hipMemcpyAsync (host to device)"); }
void HipRuntime::synchronize(void* stream) const { hip().chk(hip().StreamSynchronize(stream), "hipStreamSynchronize"); }

DeviceBuffer::DeviceBuffer(size_t bytes) : bytes_(bytes) {
  if (!bytes) return;
  p_ = HipRuntime::get().malloc_device(bytes);
  HipRuntime::get().memset_device(p_, 0, bytes);   // (null stream + device synchronise: a context's non-blocking stream does not wait for the null stream)
}
DeviceBuffer::~DeviceBuffer() { if (p_) HipRuntime::get().free_device(p_); }
DeviceBuffer& DeviceBuffer::operator=(DeviceBuffer&& o) noexcept {
  if (this != &o) {
    if (p_) HipRuntime::get().free_device(p_);
    p_ = o.p_; bytes_ = o.bytes_;
    o.p_ = nullptr; o.bytes_ = 0;
  }
  return *this;
}

MemOps hip_mem_ops() {
  return MemOps{[](void* d, const void* s, size_t n, void* st) { HipRuntime::get().copy_to_host(d, s, n, st); },
                [](void* d, const void* s, size_t n, void* st) { HipRuntime::get().copy_to_device(d, s, n, st); },
                [](void* st) { HipRuntime::get().synchronize(st); }};
}
MemOps host_mem_ops() {
  return MemOps{[](void* d, const void* s, size_t n, void*) { std::memcpy(d, s, n); }, [](void* d, const void* s, size_t n, void*) { std::memcpy(d, s, n); }, [](void*) {}};
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// RCCL (rccl.h: ncclUniqueId = 128 opaque bytes passed BY VALUE to ncclCommInitRank; ncclResult_t ncclSuccess = 0; ncclDataType_t ncclUint8 = 1)
// ---------------------------------------------------------------------------------------------------------------------------------------------
namespace {
struct NcclId { char internal[128]; };
class RcclCollective final : public Collective {
 public:
  RcclCollective(int rank, int world, int device, const std::string& dir) : rank_(rank), world_(world) {
    for (const char* n : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so", "/opt/rocm/lib/librccl.so.1"})
      if ((lib_ = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!lib_) fail("librccl.so is not loadable on this box (the all-gather of the multi-GPU path needs RCCL)");
    GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(sym(lib_, "ncclGetUniqueId", "librccl"));
    CommInitRank = reinterpret_cast<decltype(CommInitRank)>(sym(lib_, "ncclCommInitRank", "librccl"));
    AllGather = reinterpret_cast<decltype(AllGather)>(sym(lib_, "ncclAllGather", "librccl"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym(lib_, "ncclCommDestroy", "librccl"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym(lib_, "ncclGetErrorString", "librccl"));
    GetVersion = reinterpret_cast<decltype(GetVersion)>(dlsym(lib_, "ncclGetVersion"));
    HipRuntime::get().set_device(device);
    NcclId id;
    std::memset(&id, 0, sizeof(id));
    const std::string path = dir + "/nccl_unique_id.bin";
    if (rank == 0) {
      chk(GetUniqueId(&id), "ncclGetUniqueId");
      write_atomically(path, &id, sizeof(id));
    } else {
      read_when_there(path, &id, sizeof(id), 120.0);
    }
    chk(CommInitRank(&comm_, world, id, rank), "ncclCommInitRank");
  }
  ~RcclCollective() override { if (comm_) (void)CommDestroy(comm_); }
  int rank() const override { return rank_; }
  int world() const override { return world_; }
  std::string name() const override {
    int v = 0;
    if (GetVersion) (void)GetVersion(&v);
    return "rccl " + std::to_string(v / 10000) + "." + std::to_string(v / 100 % 100) + "." + std::to_string(v % 100);
  }
  void all_gather(const void* d_send, void* d_recv, size_t bytes, void* stream) override {
    chk(AllGather(d_send, d_recv, bytes, 1 /* ncclUint8 */, comm_, stream), "ncclAllGather");
  }

 private:
  void chk(int r, const char* what) const { if (r != 0) fail(std::string(what) + ": " + (GetErrorString ? GetErrorString(r) : "RCCL error")); }
  int rank_, world_;
  void* lib_ = nullptr;
  void* comm_ = nullptr;
  int (*GetUniqueId)(NcclId*) = nullptr;
  int (*CommInitRank)(void**, int, NcclId, int) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, void*) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*GetVersion)(int*) = nullptr;
};

// TEST ONLY.  Call k of rank r: drain the stream, copy the block to the host, publish it as <dir>/ag_<k>_<r>.bin, read the blocks of all ranks,
// copy the concatenation back on the stream and wait for it.  A rank removes its file of call k - 2 when it starts call k: a rank that has
// completed call k - 1 has read every other rank's file of that call, which the other rank wrote only after finishing ALL its reads of call k - 2.
class StagedCollective final : public Collective {
 public:
  StagedCollective(int rank, int world, std::string dir, MemOps ops) : rank_(rank), world_(world), dir_(std::move(dir)), ops_(ops) {}
  int rank() const override { return rank_; }
  int world() const override { return world_; }
  std::string name() const override { return "host-staged files (test only)"; }
  void all_gather(const void* d_send, void* d_recv, size_t bytes, void* stream) override {
    if (call_ >= 2) (void)std::remove(path(call_ - 2, rank_).c_str());
    std::vector<char> mine(bytes), all(bytes * (size_t)world_);
    ops_.to_host(mine.data(), d_send, bytes, stream);
    ops_.sync(stream);
    write_atomically(path(call_, rank_), mine.data(), bytes);
    for (int q = 0; q < world_; ++q) {
      if (q == rank_) std::memcpy(all.data() + (size_t)q * bytes, mine.data(), bytes);
      else read_when_there(path(call_, q), all.data() + (size_t)q * bytes, bytes, 300.0);
    }
    ops_.to_device(d_recv, all.data(), all.size(), stream);
    ops_.sync(stream);
    ++call_;
  }

 private:
  std::string path(long k, int r) const { return dir_ + "/ag_" + std::to_string(k) + "_" + std::to_string(r) + ".bin"; }
  int rank_, world_;
  std::string dir_;
  MemOps ops_;
  long call_ = 0;
};
}  // namespace

std::unique_ptr<Collective> make_rccl_collective(int rank, int world, int device, const std::string& rendezvous_dir) {
  return std::unique_ptr<Collective>(new RcclCollective(rank, world, device, rendezvous_dir));
}
std::unique_ptr<Collective> make_staged_collective(int rank, int world, const std::string& rendezvous_dir, MemOps ops) {
  return std::unique_ptr<Collective>(new StagedCollective(rank, world, rendezvous_dir, ops));
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
namespace {
void se3d_chk(se3d_ctx* c, int rc, const char* what) {
  if (rc != SE3D_OK) fail(std::string(what) + " failed (" + std::to_string(rc) + "): " + se3d_last_error(c));
}
}  // namespace

HypothesisShard::HypothesisShard(se3d_ctx* ctx, Collective& coll, int n_frames, int n_seq) : ctx_(ctx), coll_(coll), n_frames_(n_frames), n_seq_(n_seq) {
  se3d_chk(ctx_, se3d_set_shard(ctx_, coll.rank(), coll.world()), "se3d_set_shard");
  const size_t b1 = (size_t)se3d_shard_bytes(ctx_, 1, n_frames), b2 = (size_t)se3d_shard_bytes(ctx_, 2, n_seq);
  pack1_ = DeviceBuffer(b1); gath1_ = DeviceBuffer(b1 * (size_t)coll.world());
  pack2_ = DeviceBuffer(b2); gath2_ = DeviceBuffer(b2 * (size_t)coll.world());
}
void HypothesisShard::run_stage1() {
  se3d_chk(ctx_, se3d_run_stage1_shard_pre(ctx_, pack1_.data()), "se3d_run_stage1_shard_pre");
  coll_.all_gather(pack1_.data(), gath1_.data(), pack1_.bytes(), se3d_stream_handle(ctx_));   // ordered after the pack on the context's stream
  se3d_chk(ctx_, se3d_run_stage1_shard_post(ctx_, gath1_.data()), "se3d_run_stage1_shard_post");
}
void HypothesisShard::run_stage2(const double* t, int frame_slot0) {
  se3d_chk(ctx_, se3d_run_stage2_shard_pre(ctx_, n_seq_, frame_slot0, t, pack2_.data()), "se3d_run_stage2_shard_pre");
  coll_.all_gather(pack2_.data(), gath2_.data(), pack2_.bytes(), se3d_stream_handle(ctx_));
  se3d_chk(ctx_, se3d_run_stage2_shard_post(ctx_, n_seq_, gath2_.data()), "se3d_run_stage2_shard_post");
}

SkeletonExchange::SkeletonExchange(se3d_ctx* ctx, Collective& coll, int n_frames, int capacity)
    : ctx_(ctx), coll_(coll), n_frames_(n_frames), capacity_(capacity),
      pack_p_((size_t)capacity * sizeof(se3d_person3d)), pack_o_((size_t)(n_frames + 2) * sizeof(int32_t)),
      gath_p_((size_t)capacity * sizeof(se3d_person3d) * (size_t)coll.world()), gath_o_((size_t)(n_frames + 2) * sizeof(int32_t) * (size_t)coll.world()) {}
void SkeletonExchange::exchange() {
  se3d_chk(ctx_, se3d_pack_stage2_walk_device_dense(ctx_, n_frames_, capacity_, pack_p_.data(), pack_o_.data()), "se3d_pack_stage2_walk_device_dense");
  void* st = se3d_stream_handle(ctx_);
  coll_.all_gather(pack_p_.data(), gath_p_.data(), pack_p_.bytes(), st);
  coll_.all_gather(pack_o_.data(), gath_o_.data(), pack_o_.bytes(), st);
}
void SkeletonExchange::fetch(std::vector<std::vector<se3d_person3d>>& persons, std::vector<std::vector<int32_t>>& offsets) {
  const int W = coll_.world();
  void* st = se3d_stream_handle(ctx_);
  std::vector<se3d_person3d> p((size_t)capacity_ * W);
  std::vector<int32_t> o((size_t)(n_frames_ + 2) * W);
  HipRuntime::get().copy_to_host(p.data(), gath_p_.data(), gath_p_.bytes(), st);
  HipRuntime::get().copy_to_host(o.data(), gath_o_.data(), gath_o_.bytes(), st);
  HipRuntime::get().synchronize(st);
  persons.assign(W, {});
  offsets.assign(W, {});
  for (int r = 0; r < W; ++r) {
    offsets[r].assign(o.begin() + (size_t)r * (n_frames_ + 2), o.begin() + (size_t)(r + 1) * (n_frames_ + 2));
    const int n = offsets[r][n_frames_];
    persons[r].assign(p.begin() + (size_t)r * capacity_, p.begin() + (size_t)r * capacity_ + n);
  }
}

}  // namespace se3d
