// Staging kernels of the multi-GPU (row-panel sharded) step: every exchange is an all-gather of equal-sized,
// contiguous slots -- rank g packs what it owns into slot g of a send buffer, the host's collective (RCCL through
// torch.distributed, or any ekf_allgather_fn) fills the `world` slots of the receive buffer, and the other ranks'
// slots are unpacked into the library's own buffers.  HBM-bound copies, 16 bytes per lane.
#pragma once
#include <hip/hip_runtime.h>

namespace ekf {

constexpr int kMaxWorld = 16;

// Per-rank ranges (rows of Sigma / W / V, rows of S, or feature indices) of one exchange.
struct ShardTab {
  int world, self;
  int start[kMaxWorld];
  int count[kMaxWorld];
};

// rows [row0, row0 + nrows) x columns [col0, col0 + ncols) of src (row stride ld) -> dst, nrows x ncols contiguous.
// ncols, col0 and ld are multiples of 16 bytes / sizeof(T).
template <typename T>
__global__ void k_pack_rows(const T* __restrict__ src, int ld, int row0, int nrows, int col0, int ncols,
                            T* __restrict__ dst) {
  constexpr int V = 16 / sizeof(T);
  typedef T vec_t __attribute__((ext_vector_type(V)));
  const int nv = ncols / V;
  for (int r = blockIdx.y; r < nrows; r += gridDim.y) {
    const T* s = src + (size_t)(row0 + r) * ld + col0;
    T* d = dst + (size_t)r * ncols;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < nv; c += gridDim.x * blockDim.x)
      *reinterpret_cast<vec_t*>(d + (size_t)c * V) = *reinterpret_cast<const vec_t*>(s + (size_t)c * V);
  }
}

// recv = `world` slots of slot_elems scalars; slot g holds tab.count[g] rows of ncols scalars, destined for rows
// tab.start[g] .. of dst (row stride ld), columns col0 ..  The own slot is skipped (the data is already in place).
// grid: (x over columns, y over rows of a slot, z over ranks).
template <typename T>
__global__ void k_unpack_rows(const T* __restrict__ recv, size_t slot_elems, int ncols, T* __restrict__ dst, int ld,
                              int col0, ShardTab tab) {
  constexpr int V = 16 / sizeof(T);
  typedef T vec_t __attribute__((ext_vector_type(V)));
  const int g = blockIdx.z;
  if (g == tab.self) return;
  const int nv = ncols / V;
  const T* slot = recv + (size_t)g * slot_elems;
  for (int r = blockIdx.y; r < tab.count[g]; r += gridDim.y) {
    const T* s = slot + (size_t)r * ncols;
    T* d = dst + (size_t)(tab.start[g] + r) * ld + col0;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < nv; c += gridDim.x * blockDim.x)
      *reinterpret_cast<vec_t*>(d + (size_t)c * V) = *reinterpret_cast<const vec_t*>(s + (size_t)c * V);
  }
}

// ---- distributed chain (a rank owns the 128-row blocks I of S with I % world == rank): the panel of block step j --------
// own blocks of column block j -> the send slot ([block][128][128], the order of the list)
template <typename T>
__global__ void k_dist_pack_panel(const T* __restrict__ Ycol, int ld, const int* __restrict__ blocks, int nblocks,
                                  T* __restrict__ dst) {
  constexpr int V = 16 / sizeof(T);
  typedef T vec_t __attribute__((ext_vector_type(V)));
  const int row = blockIdx.x;                      // 0 .. 128 nblocks
  if (row >= 128 * nblocks) return;
  const int b = row >> 7, r = row & 127;
  const T* s = Ycol + (size_t)(blocks[b] * 128 + r) * ld;
  T* d = dst + (size_t)row * 128;
  for (int c = threadIdx.x; c < 128 / V; c += blockDim.x)
    *reinterpret_cast<vec_t*>(d + (size_t)c * V) = *reinterpret_cast<const vec_t*>(s + (size_t)c * V);
}
// slot g of recv holds rank g's blocks of this step in ascending order: I = first(g), first(g) + world, ... < nblk with
// first(g) the smallest I > j, I % world == g.  Everybody else's blocks go to their rows of the column block.
template <typename T>
__global__ void k_dist_unpack_panel(const T* __restrict__ recv, size_t slot_elems, T* __restrict__ Ycol, int ld, int j, int nblk,
                                    int world, int self) {
  constexpr int V = 16 / sizeof(T);
  typedef T vec_t __attribute__((ext_vector_type(V)));
  const int g = blockIdx.z;
  if (g == self) return;
  int first = j + 1 + ((g - (j + 1)) % world + world) % world;
  const int cnt = first < nblk ? (nblk - 1 - first) / world + 1 : 0;
  const int row = blockIdx.x;
  if (row >= 128 * cnt) return;
  const int b = row >> 7, r = row & 127;
  const T* s = recv + (size_t)g * slot_elems + (size_t)row * 128;
  T* d = Ycol + (size_t)((first + b * world) * 128 + r) * ld;
  for (int c = threadIdx.x; c < 128 / V; c += blockDim.x)
    *reinterpret_cast<vec_t*>(d + (size_t)c * V) = *reinterpret_cast<const vec_t*>(s + (size_t)c * V);
}

// Per-feature record of the "reassemble H" exchange: [h (2) | Hc (14) | Hf (12) | flag (1)] = 29 scalars.
constexpr int kFeatRec = 32;       // padded to 32 scalars: records stay 16-byte aligned

template <typename T>
__global__ void k_pack_features(const T* __restrict__ h, const T* __restrict__ Hc, const T* __restrict__ Hf,
                                const unsigned char* __restrict__ flags, int f0, int count, T* __restrict__ dst) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = t / kFeatRec, e = t % kFeatRec;
  if (i >= count) return;
  const int f = f0 + i;
  T v = T(0);
  if (e < 2) v = h[2 * f + e];
  else if (e < 16) v = Hc[(size_t)f * 14 + (e - 2)];
  else if (e < 28) v = Hf[(size_t)f * 12 + (e - 16)];
  else if (e == 28) v = T(flags[f]);
  dst[(size_t)i * kFeatRec + e] = v;
}

template <typename T>
__global__ void k_unpack_features(const T* __restrict__ recv, size_t slot_elems, T* __restrict__ h, T* __restrict__ Hc,
                                  T* __restrict__ Hf, unsigned char* __restrict__ flags, ShardTab tab) {
  const int g = blockIdx.y;
  if (g == tab.self) return;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = t / kFeatRec, e = t % kFeatRec;
  if (i >= tab.count[g]) return;
  const int f = tab.start[g] + i;
  const T v = recv[(size_t)g * slot_elems + (size_t)i * kFeatRec + e];
  if (e < 2) h[2 * f + e] = v;
  else if (e < 16) Hc[(size_t)f * 14 + (e - 2)] = v;
  else if (e < 28) Hf[(size_t)f * 12 + (e - 16)] = v;
  else if (e == 28) flags[f] = (unsigned char)v;
}

// One byte per feature (the linearity flags of convert2XYZ_ifLinearAll), carried as one scalar each.
template <typename T>
__global__ void k_pack_flags(const unsigned char* __restrict__ flags, int f0, int count, T* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) dst[i] = T(flags[f0 + i]);
}
template <typename T>
__global__ void k_unpack_flags(const T* __restrict__ recv, size_t slot_elems, unsigned char* __restrict__ flags,
                               ShardTab tab) {
  const int g = blockIdx.y;
  if (g == tab.self) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < tab.count[g]) flags[tab.start[g] + i] = (unsigned char)recv[(size_t)g * slot_elems + i];
}

// 2x2 St blocks (4 scalars per feature) of a contiguous feature range / of list positions [k0, k0 + count) of a list.
template <typename T>
__global__ void k_pack_sd(const T* __restrict__ Sd, const int* __restrict__ list, int first, int count, T* __restrict__ dst) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 4 * count) return;
  const int f = list ? list[first + t / 4] : first + t / 4;
  dst[t] = Sd[(size_t)f * 4 + (t & 3)];
}
template <typename T>
__global__ void k_unpack_sd(const T* __restrict__ recv, size_t slot_elems, const int* __restrict__ list, T* __restrict__ Sd,
                            ShardTab tab) {
  const int g = blockIdx.y;
  if (g == tab.self) return;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 4 * tab.count[g]) return;
  const int f = list ? list[tab.start[g] + t / 4] : tab.start[g] + t / 4;
  Sd[(size_t)f * 4 + (t & 3)] = recv[(size_t)g * slot_elems + t];
}

// Diagonal blocks of Sigma (the 6 x 6 -- XYZ feature: 3 x 3, zero-padded -- block of every feature of a contiguous range):
// what the map getters and the removal archive read of a feature's covariance (RosVSLAMRansac.cpp:177-183, 376-388,
// vslamRansac.cpp:394-404).  Only the owner's rows of Sigma are valid, so the owners' blocks are gathered first.
template <typename T>
__global__ void k_pack_diag(const T* __restrict__ S, int ld, const int* __restrict__ pos, const int* __restrict__ coding,
                            int first, int count, T* __restrict__ dst) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 36 * count) return;
  const int f = first + t / 36, e = t % 36, a = e / 6, b = e % 6;
  const int fs = coding[f] ? 3 : 6;
  dst[t] = (a < fs && b < fs) ? S[(size_t)(pos[f] + a) * ld + pos[f] + b] : T(0);
}
template <typename T>
__global__ void k_unpack_diag(const T* __restrict__ recv, size_t slot_elems, T* __restrict__ S, int ld,
                              const int* __restrict__ pos, const int* __restrict__ coding, ShardTab tab) {
  const int g = blockIdx.y;
  if (g == tab.self) return;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 36 * tab.count[g]) return;
  const int f = tab.start[g] + t / 36, e = t % 36, a = e / 6, b = e % 6;
  const int fs = coding[f] ? 3 : 6;
  if (a < fs && b < fs) S[(size_t)(pos[f] + a) * ld + pos[f] + b] = recv[(size_t)g * slot_elems + t];
}

// column `col` of the M x M inlier mask, rows [j0, j0 + count) -> bytes
__global__ void k_pack_mask_col(const unsigned char* __restrict__ mask, int M, int col, int j0, int count,
                                unsigned char* __restrict__ dst) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < count) dst[t] = mask[(size_t)(j0 + t) * M + col];
}

}  // namespace ekf
