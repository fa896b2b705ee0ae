// se3d_exchange.hpp — the multi-GPU exchange of the path in the C++ host layer (round 6; SURVEY 8e, north_star: "person hypotheses shard across the
// 8 GPUs of one node with an RCCL all-gather of final 3D skeletons over xGMI").  One process per GPU; the only collective of the path is an
// all-gather, enqueued on the context's own HIP stream (se3d_stream_handle) so that no host synchronisation sits between the kernels that fill the
// send buffer, the collective and the kernels that consume the result.
//
//   Collective          all-gather backends: RCCL (librccl.so, loaded with dlopen — `make -C host` needs neither its headers nor the library; a
//                       process that asks for it on a box without RCCL gets an exception) and a TEST-ONLY host-staged backend (device buffers
//                       through host memory and files of a rendezvous directory) that lets several ranks share ONE GPU, which RCCL refuses —
//                       the C++ twin of shard.HostStagedDist of the Python mirror;
//   HypothesisShard     north_star config 5: every rank holds the same frames, the association (rows N, E, G, M) and the track lifecycle (A2, L)
//                       are replicated, rank r triangulates the hypotheses h with h % world == r and refines the detections i with i % world == r:
//                       se3d_run_stage1_shard_pre -> all-gather -> _post, se3d_run_stage2_shard_pre -> all-gather -> _post (include/se3d.h);
//   SkeletonExchange    the throughput sharding (ranks own disjoint sequences): the published persons of a resident walk, packed densely on the
//                       device (se3d_pack_stage2_walk_device_dense), all-gathered so that every rank holds the final skeletons of all ranks.
// The reference has no counterpart (single process, OpenMP over hypotheses S3D:676-681 and detections PP:582-587).
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../include/se3d.h"

namespace se3d {

// The few HIP runtime entry points the exchange needs for its own buffers (libamdhip64.so through dlopen: the host layer is compiled by plain g++
// without the HIP headers).  Throws std::runtime_error when the runtime is missing or a call fails.
struct HipRuntime {
  static const HipRuntime& get();
  void* malloc_device(size_t bytes) const;
  void free_device(void* p) const;
  void set_device(int device) const;
  void memset_device(void* p, int value, size_t bytes) const;
  void copy_to_host(void* dst, const void* d_src, size_t bytes, void* stream) const;     // asynchronous on `stream`
  void copy_to_device(void* d_dst, const void* src, size_t bytes, void* stream) const;   // asynchronous on `stream`
  void synchronize(void* stream) const;
};

class DeviceBuffer {   // RAII device allocation, zero-filled (the fill is complete when the constructor returns)
 public:
  explicit DeviceBuffer(size_t bytes = 0);
  ~DeviceBuffer();
  DeviceBuffer(DeviceBuffer&& o) noexcept : p_(o.p_), bytes_(o.bytes_) { o.p_ = nullptr; o.bytes_ = 0; }
  DeviceBuffer& operator=(DeviceBuffer&& o) noexcept;
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  void* data() const { return p_; }
  size_t bytes() const { return bytes_; }

 private:
  void* p_ = nullptr;
  size_t bytes_ = 0;
};

// How a staged backend moves bytes between "device" and host memory.  The GPU form goes through HipRuntime; the CPU test of the rendezvous
// (host/se3d_exchange_test.cpp) passes plain memcpy so that the protocol runs without a GPU.
struct MemOps {
  void (*to_host)(void* dst, const void* d_src, size_t bytes, void* stream);
  void (*to_device)(void* d_dst, const void* src, size_t bytes, void* stream);
  void (*sync)(void* stream);
};
MemOps hip_mem_ops();
MemOps host_mem_ops();   // memcpy (tests)

class Collective {
 public:
  virtual ~Collective() = default;
  virtual int rank() const = 0;
  virtual int world() const = 0;
  virtual std::string name() const = 0;
  // Every rank contributes `bytes` bytes at d_send; d_recv receives world x bytes, rank r's block at offset r x bytes.  Ordered on `stream`
  // (a hipStream_t, e.g. se3d_stream_handle): after the work already enqueued there, before the work enqueued afterwards.
  virtual void all_gather(const void* d_send, void* d_recv, size_t bytes, void* stream) = 0;
};
// RCCL over xGMI (the product backend).  The ncclUniqueId travels through `rendezvous_dir` (rank 0 writes it, the others wait for the file).
std::unique_ptr<Collective> make_rccl_collective(int rank, int world, int device, const std::string& rendezvous_dir);
// TEST ONLY: device buffers staged through host memory, blocks exchanged as files of `rendezvous_dir`.  Synchronous; its speed means nothing.
std::unique_ptr<Collective> make_staged_collective(int rank, int world, const std::string& rendezvous_dir, MemOps ops = hip_mem_ops());

class HypothesisShard {
 public:
  // ctx: a context whose limits.max_frames >= n_frames; n_seq sequences per stage-2 step.  Calls se3d_set_shard(ctx, rank, world).
  HypothesisShard(se3d_ctx* ctx, Collective& coll, int n_frames, int n_seq);
  // rows N, E, G, M on every rank, rows T, P, O, U of the own hypotheses, all-gather of the [frame][ceil(H / world)] person blocks, rows X, D on every rank
  void run_stage1();
  // rows Y, A2 on every rank, rows F, R, Z of the own detections, all-gather of the [sequence][ceil(H / world)] records, row L on every rank
  void run_stage2(const double* t, int frame_slot0);
  size_t allgather_bytes_stage1() const { return pack1_.bytes() * (size_t)coll_.world(); }
  size_t allgather_bytes_stage2() const { return pack2_.bytes() * (size_t)coll_.world(); }

 private:
  se3d_ctx* ctx_;
  Collective& coll_;
  int n_frames_, n_seq_;
  DeviceBuffer pack1_, gath1_, pack2_, gath2_;
};

class SkeletonExchange {
 public:
  // n_frames result slots of the resident walk, `capacity` persons per rank and step in the dense block
  SkeletonExchange(se3d_ctx* ctx, Collective& coll, int n_frames, int capacity);
  void exchange();   // pack (device) -> all-gather of persons and offsets, on the context's stream; asynchronous
  // Host copies of the gathered blocks (waits for the stream): persons[r] = rank r's published persons back to back,
  // offsets[r] = [n_frames + 2] (per-frame start, total written, persons that did not fit)
  void fetch(std::vector<std::vector<se3d_person3d>>& persons, std::vector<std::vector<int32_t>>& offsets);

 private:
  se3d_ctx* ctx_;
  Collective& coll_;
  int n_frames_, capacity_;
  DeviceBuffer pack_p_, pack_o_, gath_p_, gath_o_;
};

}  // namespace se3d
