// mobi_gop.hip -- the two small kernels around the frame-parallel parse (mobi_gop.h says what and why): one lane per CLIP, walking its K
// frames in order.  Everything heavy -- the parse of the n * K virtual clips -- is the unchanged parse kernels' (mobi_lsparse.hip,
// mobi_dparse.hip); these two carry the few bytes of decoder state that chain from frame to frame (MD.cs:113-154, 224-236, 3884-3925).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "mobi_gop.h"
#include "mobi_kernels.h"

// start states of every virtual clip: frame 0 starts from the batch's state ring, frame k + 1 from frame k's start state and frame k's header
extern "C" __global__ __launch_bounds__(64) void mobi_gop_prepare(MobiGopArgs A) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= A.n) return;
  if (A.P.bit_len[c] == MOBI_DP_SKIP) return; // the host parser's clip (all K entries say so)
  MobiDevState g = A.ring_in[c];
  MobiDevState *sin = (MobiDevState *)A.P.state_in;
  sin[c] = g;
  const int moflex = A.P.version == 2 /* MOBI_VERSION_MOFLEX3DS */;
  for (int k = 1; k < A.K; k++) {
    const size_t v = (size_t)(k - 1) * A.n + c;
    mobi_gop_next_guess(moflex, A.P.bits + A.P.bit_off[v], A.P.bit_len[v], g); // (the staging area carries 32 zero bytes behind every frame)
    sin[v + A.n] = g;
  }
}

// Behind the parse kernels: frame by frame, was the start state the true one?  Then what the frame left, merged with what it did not touch,
// is the true state behind it, and its tail follows from its command list and the tail before (mobi_state.h).  The first frame of a clip that
// a device parser did not finish, or that started from a wrong prediction, ends the clip's chain: the host parser takes that frame and the
// ones behind it over, from state_in[v] -- which this kernel overwrites with the TRUE start state -- and the tail before it.
extern "C" __global__ __launch_bounds__(64) void mobi_gop_chain(MobiGopArgs A) {
  __shared__ uint8_t izz[80];
  if (threadIdx.x < 64) izz[A.P.tables[MOBI_DT_ZZ8 + threadIdx.x]] = (uint8_t)threadIdx.x;
  if (threadIdx.x < 16) izz[64 + A.P.tables[MOBI_DT_ZZ4 + threadIdx.x]] = (uint8_t)threadIdx.x;
  __syncthreads();
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= A.n) return;
  if (A.P.bit_len[c] == MOBI_DP_SKIP) return;
  MobiDevState *sin = (MobiDevState *)A.P.state_in;
  const int n_mbs = A.P.mbw * A.P.mbh;
  MobiDevState cur = A.ring_in[c];
  const MobiDevTail *tprev = A.rtail_in + c;
  for (int k = 0; k < A.K; k++) {
    const size_t v = (size_t)k * A.n + c;
    MobiDevResult *r = A.P.res + v;
    if (k > 0 && !mobi_gop_guess_ok(sin[v], cur)) {
      r->rc = MOBI_GOP_RC_CHAIN;
      sin[v] = cur;
      return;
    }
    sin[v] = cur; // (equal in everything a parse reads; the carried bytes are now the true ones: a hand-over starts here)
    if (r->rc != 0) return;
    const bool iframe = r->frame_type == 1;
    MobiDevState out = A.P.state_out[v];
    mobi_gop_merge(cur, iframe, out);
    A.P.state_out[v] = out;
    cur = out;
    // the tail (Internal[90..217], the MV row cache): as mobi_parse_tail, with the tail of the frame before as its input
    const MbDesc *desc = A.P.desc + v * n_mbs;
    const uint32_t *pay = A.P.payload + v * A.P.pay_cap; // (pay_local: MbDesc.payload_off counts from the virtual clip's own part)
    MobiTailScan sc;
    mobi_tail_scan_init(sc);
    for (int mb = n_mbs - 1; mb >= 0 && !sc.done; mb--) {
      const uint4 d = *(const uint4 *)(desc + mb);
      const int nw = (int)(d.z & 0x3FF);
      if (!nw) continue;
      const bool intra = (d.y & 1) == MOBI_MB_INTRA;
      const uint32_t nl = (d.y >> 1) & 0x7F, dual = (d.y >> 26) & 3;
      const uint32_t woff = d.x + (intra ? MOBI_INTRA_RECORDS : (nl > 1 && !dual) ? MOBI_MV_CELLS : 0);
      mobi_tail_scan_mb(sc, pay + woff, nw, woff, (d.y >> 14) & 0x3F, izz, izz + 64);
    }
    MobiDevTail *tout = A.P.tail_out + v;
    mobi_tail_finish(sc, pay, A.P.scale + (size_t)(cur.quant & 63) * MOBI_SCALE_STRIDE, *tprev, *tout);
    if (iframe) // an I-frame does not touch the MV row cache (MD.cs:224-249); a P-frame's was written by the parse kernels
      for (int i = 0; i < 2 * (A.P.mbw + 2); i++) tout->mvc[i] = tprev->mvc[i];
    tprev = tout;
  }
  A.ring_out[c] = cur;
  A.rtail_out[c] = *tprev;
}

// ---- wavefront-sorted intra launch items of a group's frames (mobi_gop.h) --------------------------------------------------------------
// A workgroup = 64 virtual clips of one frame, a wave per 16 of them, lanes over a clip's list.  Counts go through LDS: one global atomic per
// workgroup and wavefront, not one per item (all of a frame's 24576 x 100 items land on a few hundred counters: 9 ms per group of five
// frames that way, measured).
#define SORT_WG_CLIPS 64
__device__ __forceinline__ void sort_count(const MobiGopSortArgs &A, int k, int c0, uint32_t *cnt) {
  for (int l = threadIdx.x; l < MOBI_SORT_LEVELS; l += 256) cnt[l] = 0;
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int j = wave; j < SORT_WG_CLIPS && c0 + j < A.n; j += 4) {
    const size_t v = (size_t)k * A.n + c0 + j;
    const uint32_t ni = A.res[v].n_intra;
    const uint32_t *items = A.items + v * A.n_mbs;
    for (uint32_t i = lane; i < ni; i += 64) {
      const uint32_t mb = items[i] & 0x1FFFu, my = mb / (uint32_t)A.mbw;
      atomicAdd(&cnt[(mb - my * A.mbw) + 2 * my], 1u);
    }
  }
  __syncthreads();
}
extern "C" __global__ __launch_bounds__(256) void mobi_gop_fronts(MobiGopSortArgs A) {
  __shared__ uint32_t cnt[MOBI_SORT_LEVELS];
  const int k = blockIdx.y, c0 = blockIdx.x * SORT_WG_CLIPS;
  sort_count(A, k, c0, cnt);
  for (int l = threadIdx.x; l < MOBI_SORT_LEVELS; l += 256)
    if (cnt[l]) atomicAdd(&A.hist[k * MOBI_SORT_LEVELS + l], cnt[l]);
}
// one workgroup per frame: where each wavefront starts (on a wave of four items)
extern "C" __global__ __launch_bounds__(64) void mobi_gop_front_starts(MobiGopSortArgs A) {
  const int k = blockIdx.x, lane = threadIdx.x;
  constexpr int PER = MOBI_SORT_LEVELS / 64;
  uint32_t len[PER], mine = 0;
#pragma unroll
  for (int j = 0; j < PER; j++) { len[j] = (A.hist[k * MOBI_SORT_LEVELS + lane * PER + j] + 3u) & ~3u; mine += len[j]; }
  uint32_t at = mine; // inclusive prefix over the lanes
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(at, d); if (lane >= d) at += o; }
  at -= mine;
#pragma unroll
  for (int j = 0; j < PER; j++) { A.start[k * MOBI_SORT_LEVELS + lane * PER + j] = at; at += len[j]; }
}
// the workgroups of mobi_gop_fronts again: room for the workgroup's items in every wavefront's run, then the items into it
extern "C" __global__ __launch_bounds__(256) void mobi_gop_scatter(MobiGopSortArgs A) {
  __shared__ uint32_t cnt[MOBI_SORT_LEVELS], base[MOBI_SORT_LEVELS];
  const int k = blockIdx.y, c0 = blockIdx.x * SORT_WG_CLIPS;
  sort_count(A, k, c0, cnt);
  for (int l = threadIdx.x; l < MOBI_SORT_LEVELS; l += 256) {
    base[l] = cnt[l] ? A.start[k * MOBI_SORT_LEVELS + l] + atomicAdd(&A.cursor[k * MOBI_SORT_LEVELS + l], cnt[l]) : 0u;
    cnt[l] = 0;
  }
  __syncthreads();
  uint4 *out = (uint4 *)(A.sorted + A.sorted_off[k] * 4);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int j = wave; j < SORT_WG_CLIPS && c0 + j < A.n; j += 4) {
    const size_t v = (size_t)k * A.n + c0 + j;
    const uint32_t ni = A.res[v].n_intra;
    const MbDesc *desc = A.desc + v * A.n_mbs;
    const uint32_t *items = A.items + v * A.n_mbs;
    for (uint32_t i = lane; i < ni; i += 64) {
      const uint32_t mb = items[i] & 0x1FFFu, my = mb / (uint32_t)A.mbw, l = (mb - my * A.mbw) + 2 * my;
      const uint4 d = *(const uint4 *)(desc + mb); // payload_off, w1, w2, w3
      const uint32_t pos = base[l] + atomicAdd(&cnt[l], 1u);
      if (pos < A.sorted_cap[k]) // (always: the caller sized the room from the same counts)
        out[pos] = uint4{MOBI_ITEM((uint32_t)(c0 + j), mb), d.y, d.x, (d.w & (0xFFFF0007u | MOBI_W3_WIDE)) | ((d.z & 0x3FFu) << 5)};
    }
  }
}
// padding rows: MOBI_ITEM_NONE and nothing else set (no record words to fetch, nobody to wait for)
extern "C" __global__ __launch_bounds__(256) void mobi_gop_sort_fill(uint4 *out, size_t n_items) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_items; i += (size_t)gridDim.x * 256) out[i] = uint4{MOBI_ITEM_NONE, 0u, 0u, 0u};
}
extern "C" int mobi_launch_gop_sort(const MobiGopSortArgs *a, hipStream_t s) {
  if (a->n <= 0 || a->K <= 0) return 0;
  const dim3 g((unsigned)((a->n + SORT_WG_CLIPS - 1) / SORT_WG_CLIPS), (unsigned)a->K);
  const size_t total = a->sorted_off[a->K - 1] + a->sorted_cap[a->K - 1];
  if (!total) return 0;
  hipLaunchKernelGGL(mobi_gop_sort_fill, dim3((unsigned)std::min<size_t>((total + 255) / 256, 8192)), dim3(256), 0, s, (uint4 *)a->sorted, total);
  hipLaunchKernelGGL(mobi_gop_fronts, g, dim3(256), 0, s, *a);
  hipLaunchKernelGGL(mobi_gop_front_starts, dim3((unsigned)a->K), dim3(64), 0, s, *a);
  hipLaunchKernelGGL(mobi_gop_scatter, g, dim3(256), 0, s, *a);
  return (int)hipGetLastError();
}

extern "C" int mobi_launch_gop_prepare(const MobiGopArgs *a, hipStream_t s) {
  if (a->n <= 0) return 0;
  hipLaunchKernelGGL(mobi_gop_prepare, dim3((unsigned)((a->n + 63) / 64)), dim3(64), 0, s, *a);
  return (int)hipGetLastError();
}
extern "C" int mobi_launch_gop_chain(const MobiGopArgs *a, hipStream_t s) {
  if (a->n <= 0) return 0;
  hipLaunchKernelGGL(mobi_gop_chain, dim3((unsigned)((a->n + 63) / 64)), dim3(64), 0, s, *a);
  return (int)hipGetLastError();
}
