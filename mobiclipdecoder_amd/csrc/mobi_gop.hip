// mobi_gop.hip -- the two small kernels around the frame-parallel parse (mobi_gop.h says what and why): one lane per CLIP, walking its K
// frames in order.  Everything heavy -- the parse of the n * K virtual clips -- is the unchanged parse kernels' (mobi_lsparse.hip,
// mobi_dparse.hip); these two carry the few bytes of decoder state that chain from frame to frame (MD.cs:113-154, 224-236, 3884-3925).
#include <hip/hip_runtime.h>

#include "mobi_gop.h"

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
