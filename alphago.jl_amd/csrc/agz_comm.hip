// agz_comm.hip -- the ONE exchange step of the path, behind the C ABI: an RCCL all-gather over
// xGMI of the finished (moves, pi, q, z) records of every rank into every rank's device replay arena, plus the
// optional weight broadcast from the training rank.  Self-play itself has no collective (games are independent,
// /root/reference/src/train.jl:56-57); the caller this serves is the replay-buffer fill of train.jl:56-66.
//
// RCCL is bound at run time (dlopen of librccl.so.1, no link-time dependency): libagz.so must load and export
// its symbols on a host without RCCL/GPUs, and a process that already carries a copy of RCCL (a PyTorch wheel
// bundles its own) must end up with ONE instance -- dlopen by soname returns that one.
//
// Shape of the exchange (records are variable-length, ranks finish different numbers of games):
//   1. every rank packs its finished records into one device buffer            (k_pack_records)
//   2. ncclAllGather of {records, bytes} per rank                               (16 B per rank)
//   3. ncclAllGather of the payload padded to the largest rank                  (device -> device)
//   4. one wave per rank walks that rank's chunk to index the records           (k_index_records)
//   5. the chunks are compacted into the arena; only the index (40 B per game) visits the host.
// With 7 x 153 GB/s xGMI links per GPU a BASELINE configs[2] generation (8192 games, ~0.25 GB) is a
// sub-millisecond transfer: one large collective, not one per game.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>

#include "agz_engine.h"

namespace agz {

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};

static Rccl& rccl() {
  static Rccl R;
  static std::once_flag once;
  std::call_once(once, [] {
    // AGZ_RCCL_SONAME overrides the search (a site with its own RCCL build; the CPU suite uses it to force the
    // not-found path, which must be an AGZ_RCCL_ERROR status and never a crash)
    const char* forced = getenv("AGZ_RCCL_SONAME");
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    std::string why = "?";
    for (const char* n : names) {
      if (forced) n = forced;
      R.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (R.lib) break;
      const char* e = dlerror();          // (dlerror() clears the message: read it exactly once per failure)
      if (e) why = e;
      if (forced) break;
    }
    if (!R.lib) {
      R.error = std::string(forced ? forced : "librccl.so.1") + " not found: " + why;
      return;
    }
    auto sym = [&](const char* s) {
      void* p = dlsym(R.lib, s);
      if (!p && R.error.empty()) R.error = std::string("RCCL symbol missing: ") + s;
      return p;
    };
    R.GetUniqueId = (decltype(R.GetUniqueId))sym("ncclGetUniqueId");
    R.CommInitRank = (decltype(R.CommInitRank))sym("ncclCommInitRank");
    R.CommDestroy = (decltype(R.CommDestroy))sym("ncclCommDestroy");
    R.AllGather = (decltype(R.AllGather))sym("ncclAllGather");
    R.Broadcast = (decltype(R.Broadcast))sym("ncclBroadcast");
    R.GetErrorString = (decltype(R.GetErrorString))sym("ncclGetErrorString");
  });
  AGZ_REQUIRE(R.error.empty(), AGZ_RCCL_ERROR, "%s", R.error.c_str());
  return R;
}

#define AGZ_RCCL(expr)                                                                                    \
  do {                                                                                                    \
    ncclResult_t _r = (expr);                                                                             \
    if (_r != ncclSuccess)                                                                                \
      throw ::agz::Error(AGZ_RCCL_ERROR, ::agz::fmt("%s failed: %s (%s:%d)", #expr,                       \
                                                    rccl().GetErrorString ? rccl().GetErrorString(_r) : "?", \
                                                    __FILE__, __LINE__));                                 \
  } while (0)

struct Comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  Engine* engine = nullptr;
  DevBuf<int64_t> d_counts;     // [1 + world][2]: mine, then everybody's {records, bytes}
  DevBuf<uint8_t> d_pack, d_recv;   // this rank's packed records (send) / everybody's chunks (receive)
};

void comm_unique_id(uint8_t* out) {
  static_assert(AGZ_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "agz.h and rccl.h disagree on the id size");
  ncclUniqueId id;
  AGZ_RCCL(rccl().GetUniqueId(&id));
  std::memcpy(out, id.internal, NCCL_UNIQUE_ID_BYTES);
}

Comm* comm_create(Engine& E, int rank, int world, const uint8_t* idbytes) {
  AGZ_REQUIRE(world >= 1 && rank >= 0 && rank < world, AGZ_BAD_ARGUMENT, "rank %d of %d", rank, world);
  AGZ_REQUIRE(idbytes != nullptr, AGZ_BAD_ARGUMENT, "null unique id");
  AGZ_HIP(hipSetDevice(E.config().device));
  ncclUniqueId id;
  std::memcpy(id.internal, idbytes, NCCL_UNIQUE_ID_BYTES);
  std::unique_ptr<Comm> c(new Comm);
  c->rank = rank;
  c->world = world;
  c->engine = &E;
  AGZ_RCCL(rccl().CommInitRank(&c->comm, world, id, rank));
  c->d_counts.alloc((size_t)2 * (1 + world));
  return c.release();
}

void comm_destroy(Comm* c) {
  if (!c) return;
  if (c->comm) (void)rccl().CommDestroy(c->comm);
  delete c;
}

// The host logic between the two collectives, independent of what carries the bytes (RCCL here, the host's own
// library through agz_gather_plan + agz_replay_ingest_gathered): counts = {records, bytes} per rank as gathered.
// A rank that failed BEFORE the exchange still takes part in the count collective and announces {-1, its status},
// so that every rank fails the call together instead of hanging in the payload collective.
int64_t gather_plan(const int64_t* counts, int world, int64_t* total_records) {
  AGZ_REQUIRE(counts != nullptr && world >= 1, AGZ_BAD_ARGUMENT, "gather plan over %d ranks", world);
  int64_t mx = 0, total = 0;
  for (int r = 0; r < world; ++r) {
    const int64_t n = counts[2 * r], b = counts[2 * r + 1];
    AGZ_REQUIRE(n != -1, AGZ_RCCL_ERROR, "rank %d failed before the exchange (status %lld): no rank ingests anything", r,
                (long long)b);
    // a record is at least a header and at most header + max_game_length moves: bytes and records must be consistent
    AGZ_REQUIRE(n >= 0 && b >= 0 && b % 8 == 0 && (n == 0) == (b == 0) && b >= n * (int64_t)sizeof(agz_game_header),
                AGZ_RCCL_ERROR, "rank %d announced %lld records in %lld bytes", r, (long long)n, (long long)b);
    mx = std::max(mx, b);
    total += n;
  }
  if (total_records) *total_records = total;
  return (mx + 255) & ~(int64_t)255;          // chunk stride: the largest rank, padded to 256 B
}

// records of every rank -> this rank's replay arena, rank order; returns the number of games added.  Records the
// engine has already filed through this call are not sent again (Engine::records_exchanged watermark).
int64_t comm_allgather_records(Engine& E, Comm* c) {
  if (!c) return E.replay_ingest_local();     // no communicator: a single-GPU run files its own records
  AGZ_REQUIRE(c->engine == &E, AGZ_BAD_ARGUMENT, "communicator belongs to another engine");
  hipStream_t s = E.stream();
  const int W = c->world;
  // 1. everything that can fail on ONE rank happens before the first collective: this rank's unsent records are
  //    packed into its own send buffer now, and a failure is announced as {-1, status} in the count collective, so
  //    that every rank leaves the call together (a pack that failed between the two collectives left the
  //    other ranks waiting in the payload all-gather)
  const int64_t first = E.records_exchanged();
  int64_t mine[2] = {0, 0};
  std::string my_error;
  try {
    mine[0] = std::max<int64_t>(0, E.records_count() - first);
    mine[1] = E.records_packed_size(first);
    if (mine[0] > 0) {
      c->d_pack.ensure((size_t)((mine[1] + 255) & ~(int64_t)255));
      int64_t nb = 0;
      const int64_t nrec = E.pack_records_device(c->d_pack.p, (int64_t)c->d_pack.n, &nb, first);
      AGZ_REQUIRE(nrec == mine[0] && nb == mine[1], AGZ_RCCL_ERROR, "records changed during the exchange");
    }
  } catch (const Error& x) {
    mine[0] = -1;
    mine[1] = x.status;
    my_error = x.what();
  }
  // 2. counts
  AGZ_HIP(hipMemcpyAsync(c->d_counts.p, mine, sizeof(mine), hipMemcpyHostToDevice, s));
  AGZ_RCCL(rccl().AllGather(c->d_counts.p, c->d_counts.p + 2, 2, ncclInt64, c->comm, s));
  std::vector<int64_t> counts((size_t)2 * W);
  AGZ_HIP(hipMemcpyAsync(counts.data(), c->d_counts.p + 2, sizeof(int64_t) * counts.size(), hipMemcpyDeviceToHost, s));
  AGZ_HIP(hipStreamSynchronize(s));
  int64_t total = 0;
  // (a rank that failed reports AGZ_RCCL_ERROR like its peers -- the plan names it -- with its own reason appended)
  int64_t stride = 0;
  try {
    stride = gather_plan(counts.data(), W, &total);
  } catch (const Error& x) {
    if (mine[0] == -1) throw Error(x.status, std::string(x.what()) + "; this rank: " + my_error);
    throw;
  }
  if (total == 0) return 0;
  // 3. payload, padded to the largest rank (the pad bytes are never read: step 4 stops at each rank's count).  The
  //    send buffer must be readable up to the stride: grow it around what was packed if another rank's chunk is larger
  if (c->d_pack.n < (size_t)stride) {
    DevBuf<uint8_t> bigger;
    bigger.alloc((size_t)stride);
    if (mine[1] > 0) AGZ_HIP(hipMemcpyAsync(bigger.p, c->d_pack.p, (size_t)mine[1], hipMemcpyDeviceToDevice, s));
    AGZ_HIP(hipStreamSynchronize(s));
    std::swap(bigger.p, c->d_pack.p);
    std::swap(bigger.n, c->d_pack.n);
  }
  c->d_recv.ensure((size_t)stride * W);        // grow-only; nothing of an earlier call is kept in it
  AGZ_RCCL(rccl().AllGather(c->d_pack.p, c->d_recv.p, (size_t)stride, ncclUint8, c->comm, s));
  AGZ_HIP(hipStreamSynchronize(s));            // the collective has completed on this rank: its records are everybody's
  // the watermark moves with the collective, not with the local ingest: a rank whose ingest fails must not send the
  // same games again (its peers have filed them)
  E.records_mark_exchanged(first + mine[0]);
  // 4 + 5. index on the device, compact into the arena
  std::vector<int64_t> coff((size_t)W), cbytes((size_t)W), cnrec((size_t)W);
  for (int r = 0; r < W; ++r) {
    coff[r] = (int64_t)r * stride;
    cnrec[r] = counts[2 * r];
    cbytes[r] = counts[2 * r + 1];
  }
  return E.replay_ingest_chunks(c->d_recv.p, coff, cbytes, cnrec);
}

// rank `root`'s parameters overwrite every other rank's replica: ONE ncclBroadcast of the network's device master
// (Net::flux_device: every parameter as one flat f32 array, 12-24 M floats) in place, device to device.  The receivers'
// inference images are rebuilt from it by kernels before their next forward; nothing is staged through the host
// (round 4 went H2D -> broadcast -> D2H -> host repack: 476 ms at 19x19 / tower 20).
int64_t comm_broadcast_weights(Engine& E, Comm* c, int root) {
  AGZ_REQUIRE(c != nullptr, AGZ_BAD_ARGUMENT, "null communicator");
  AGZ_REQUIRE(c->engine == &E, AGZ_BAD_ARGUMENT, "communicator belongs to another engine");
  AGZ_REQUIRE(root >= 0 && root < c->world, AGZ_BAD_ARGUMENT, "root %d of %d", root, c->world);
  hipStream_t s = E.stream();
  Net& net = E.net();
  AGZ_RCCL(rccl().Broadcast(net.flux_device(), net.flux_device(), net.flux_count(), ncclFloat32, root, c->comm, s));
  if (c->rank != root) net.device_master_written();
  AGZ_HIP(hipStreamSynchronize(s));
  return (int64_t)net.flux_count();
}

}  // namespace agz
