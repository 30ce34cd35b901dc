// mobi_lsparse.hip -- the lock-step bitstream parser on gfx950: a few clips per wave, one per lane (mobi_lsparse.h has the state machine and
// says why).
//
//   mobi_parse_frames_ls   one wave = `ls_clips` clips (the other lanes idle), four or eight waves per workgroup (mobi_launch_parse_ls).  Every lane walks its own frame with
//                          ls_round(); the wave runs until the last one is done.  A lane that meets anything out of the ordinary bails out and
//                          leaves its clip to mobi_parse_frames (which, r05, leaves what it cannot finish to the host parser: mobi_abi.cpp).
//   mobi_ls_deps           one lane per intra macroblock of the clips the first kernel finished: the dependency lists (MbDesc.w4..w7).
//   mobi_parse_frames      (mobi_dparse.hip) runs afterwards, one wave per clip as always: a finished clip's wave only moves the new decoder
//                          state from its shadow copy into place; every other clip is parsed as if the first kernel had not run.
//
// LDS per workgroup: ONE copy of the table blob (18 KB) for its waves, and per wave and lane the intra records (24 words; the
// partition-tree stack of an inter macroblock lives in the same words, r06), the mode cache (40 bytes), a 128-byte ring of bitstream --
// 264 bytes (r05: 664) -- and, when the launch has room (mobi_launch_parse_ls), the motion-vector row cache (mbw + 2 words: a vector as two
// int16; else its working set is in registers and the rest in the clip's tail in HBM: mobi_lsparse.h, LsLane) -- all
// lane-interleaved (element i of lane l at i * ls_clips + l), so that the lanes reading "their" element i hit different banks.
// How many clips a wave carries is a launch argument (mobi_launch_parse_ls, with the measurements): a wave's life grows with the number of
// DIFFERENT clips it holds -- it runs until its slowest lane is done, a round costs what its lanes' different states need -- and alone on a
// SIMD it issues an instruction every ~20 clocks, so two or three short-lived waves per SIMD beat one long-lived one (r04: 32 clips per
// wave, a table copy per wave, one wave per SIMD: 31 ms per P-frame step of 24576 clips; r05: 12 per wave: 26 ms).
//
// The bitstream reaches the ring through registers, 32 bytes per lane every LS_SERVICE rounds, committed one service later: the load has
// that long to arrive, nobody waits for it.  A lane whose ring holds less than a round can ask for (LS_ROUND_BYTES) sits the round out.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "mobi_dparse.h"
#include "mobi_kernels.h"
#include "mobi_lsparse.h"

namespace {
#ifndef LS_SERVICE_N
#define LS_SERVICE_N 4
#endif
enum { LS_SERVICE = LS_SERVICE_N, LS_RING_WORDS = LS_RING / 4 }; // every LS_SERVICE rounds a lane with 32 bytes of room in its ring asks for 32 more

struct DevStore { // L = clips per wave: the per-lane state is interleaved at that stride (element i of lane l at i * L + l)
  // The MV row cache (Internal[221..]) beyond the registers that hold its working set (mobi_lsparse.h): in HBM -- this clip's MobiDevTail.mvc, two
  // int32 per entry: what a P-frame leaves IS the tail -- or, when the launch has LDS to spare (mobi_launch_parse_ls), in LDS like everything
  // else (mvl != nullptr, wave-uniform; written out at the end): one macroblock's word costs a round trip either way, but a shorter one
  int32_t *mvg;
  uint32_t *mvl;
  uint32_t *stk_, *rec_, *ring_;
  uint8_t *mc_;
  int lane, L;
  __device__ __forceinline__ uint32_t mvp_load(int i) const {
    if (mvl) return mvl[i * L + lane];
    const int2 v = ((const int2 *)mvg)[i];
    return ((uint32_t)v.x & 0xFFFFu) | ((uint32_t)v.y << 16);
  }
  __device__ __forceinline__ void mvp_store(int i, uint32_t v) {
    if (mvl) mvl[i * L + lane] = v;
    else ((int2 *)mvg)[i] = int2{(int)(int16_t)(v & 0xFFFFu), (int)(int16_t)(v >> 16)};
  }
  __device__ __forceinline__ uint32_t &stk(int i) { return stk_[i * L + lane]; }
  __device__ __forceinline__ uint32_t &rec(int i) { return rec_[i * L + lane]; }
  __device__ __forceinline__ uint8_t &mc(int i) { return mc_[i * L + lane]; }
  __device__ __forceinline__ uint32_t ring32(uint32_t rd) const { return ring_[((rd >> 2) & (LS_RING_WORDS - 1)) * L + lane]; }
};

// 16 bytes of the stream at byte offset o (a multiple of 16), bytes at and beyond len2 read as zero (the reference never reads a word
// that is not whole, MD.cs:2978-2990; the staging area carries 32 zero bytes behind every clip, so the load itself stays inside it)
__device__ __forceinline__ uint4 ls_chunk(const uint8_t *base, uint32_t o, uint32_t len2) {
  uint4 v = uint4{0, 0, 0, 0};
  if (o < len2) {
    v = *(const uint4 *)(base + o);
    if (o + 16 > len2) {
      uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int nb = (int)len2 - (int)(o + 4 * j);
        w[j] = nb >= 4 ? w[j] : nb <= 0 ? 0u : (w[j] & ((1u << (8 * nb)) - 1u));
      }
      v = uint4{w[0], w[1], w[2], w[3]};
    }
  }
  return v;
}
} // namespace

#ifndef LS_INTRA_PERIOD
#define LS_INTRA_PERIOD 3
#endif
#ifndef MOBI_LS_WAVES
#define MOBI_LS_WAVES 4 // waves per workgroup: they share one copy of the table blob in LDS (18 KB), everything else is a wave's own
#endif             // (the launch's choice: 4, or 8 where two workgroups of four would not fit a CU's LDS -- mobi_launch_parse_ls)
extern "C" __global__ __launch_bounds__(512) void mobi_parse_frames_ls(MobiDevParseArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int LS_CLIPS = A.ls_clips; // clips per wave (mobi_launch_parse_ls picks it: the launch is fastest with about two waves per SIMD)
  // A wave of this kernel is one long chain of dependent instructions and the launch is as long as that chain.  When the reconstruction of
  // the step before runs beside it (asynchronous steps: four of its waves on the same SIMD), the chain must not queue behind them.
  __builtin_amdgcn_s_setprio(3);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint8_t *tab = lds;
  DevStore m;
  m.L = LS_CLIPS;
  m.mvg = nullptr;
  const int mvp_words = A.ls_mv_lds ? A.mbw + 2 : 0;
  // (r06: the partition-tree stack lives in the intra records' words -- a macroblock is intra only by its ROOT node's code (mobi_lsparse.h,
  // LS_NODE: any deeper "intra" code ends the lane), when the stack is empty, and the records are zeroed behind that: never both at once)
  m.rec_ = (uint32_t *)(lds + MOBI_DT_BYTES + (size_t)wave * LS_CLIPS * (4 * mvp_words + 4 * MOBI_INTRA_RECORDS + LS_RING + 40));
  m.stk_ = m.rec_;
  m.ring_ = m.rec_ + MOBI_INTRA_RECORDS * LS_CLIPS;
  m.mc_ = (uint8_t *)(m.ring_ + LS_RING_WORDS * LS_CLIPS);
  m.mvl = mvp_words ? (uint32_t *)(m.mc_ + 40 * LS_CLIPS) : nullptr;
  m.lane = lane;
  for (int i = threadIdx.x; i < MOBI_DT_BYTES / 16; i += (int)blockDim.x) ((uint4 *)tab)[i] = ((const uint4 *)A.tables)[i];
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += (int)blockDim.x) ls_prepare_tables(tab, i);
  __syncthreads();

  const int clip = (int)(blockIdx.x * (blockDim.x >> 6) + wave) * LS_CLIPS + lane;
  const bool live = lane < LS_CLIPS && clip < A.n_clips && A.bit_len[clip < A.n_clips ? clip : 0] != MOBI_DP_SKIP; // (not the host parser's clips)
  const int n_mbs = A.mbw * A.mbh;
  LsCtx c;
  c.T = tab;
  c.width = A.width; c.height = A.height; c.stride = A.stride; c.lg = A.lg; c.mbw = A.mbw; c.mbh = A.mbh; c.n_mbs = n_mbs;
  c.version = A.version;
  c.pay_cap = A.pay_cap;

  LsLane s;
  const uint8_t *base = A.bits;
  uint32_t len = 0, len2 = 0, wr = 0;
  s.st = LS_DONE; s.bail = 0; s.rd = 0; s.cbits = 0; s.n_items = 0; s.pay_pos = 0; s.iframe = 0; s.quant = 0; s.yuvfmt = 0;
  if (live) {
    const MobiDevState *st = A.state_in + clip;
    s.quant = st->quant; s.yuvfmt = st->yuvfmt; s.tables_set = st->tables_set;
    s.predx = st->predx; s.predy = st->predy; // (Internal[219], [220]: an I-frame leaves them as they are)
    s.frames_started = st->frames_started + 1; // the ring turns before anything can throw (MD.cs:102-108)
    for (int i = 0; i < 40; i++) m.mc(i) = st->mcache[i];
    s.desc = A.desc + (size_t)clip * n_mbs;
    s.pay = A.payload + (A.pay_local ? (size_t)clip * A.pay_cap : (size_t)0);
    s.pay_base = A.pay_local ? 0u : (uint32_t)clip * A.pay_cap;
    s.items = A.items + (size_t)clip * n_mbs;
    s.clip = (uint32_t)(A.clip_mod ? clip % A.clip_mod : clip);
    m.mvg = A.tail_out[clip].mvc;
    base = A.bits + A.bit_off[clip];
    len = A.bit_len[clip];
    len2 = len & ~1u;
    // the ring's first LS_RING bytes
#pragma unroll
    for (int k = 0; k < LS_RING / 16; k++) {
      const uint4 v = ls_chunk(base, 16u * k, len2);
      m.ring_[(4 * k + 0) * LS_CLIPS + lane] = v.x; m.ring_[(4 * k + 1) * LS_CLIPS + lane] = v.y;
      m.ring_[(4 * k + 2) * LS_CLIPS + lane] = v.z; m.ring_[(4 * k + 3) * LS_CLIPS + lane] = v.w;
    }
    wr = LS_RING;
    ls_begin_frame(s, m, c, len);
  }
  uint4 pend0 = uint4{0, 0, 0, 0}, pend1 = uint4{0, 0, 0, 0};
  bool pending = false;
  // The intra part of the walk is half of a round's instructions for one macroblock in twenty of a P-frame, and a full wave has a lane in it
  // nine rounds out of ten: a wave of P-frames runs it every LS_INTRA_PERIOD-th round only -- its lanes wait, and each run serves three times
  // as many (simulated, tools/exp_lssched.py at 60 lanes: 10.9 -> 9.3 M instructions per P-frame).  A wave that holds an I-frame keeps every round.
  const bool intra_every_round = __builtin_amdgcn_ballot_w64(live && s.iframe) != 0;
  for (uint32_t round = 0;; round++) {
    if ((round & (LS_SERVICE - 1)) == 0) {
      if (pending) { // what the last service asked for has had LS_SERVICE rounds to arrive
        const uint32_t k = (wr >> 2) & (LS_RING_WORDS - 1); // (a multiple of 8)
        m.ring_[(k + 0) * LS_CLIPS + lane] = pend0.x; m.ring_[(k + 1) * LS_CLIPS + lane] = pend0.y;
        m.ring_[(k + 2) * LS_CLIPS + lane] = pend0.z; m.ring_[(k + 3) * LS_CLIPS + lane] = pend0.w;
        m.ring_[(k + 4) * LS_CLIPS + lane] = pend1.x; m.ring_[(k + 5) * LS_CLIPS + lane] = pend1.y;
        m.ring_[(k + 6) * LS_CLIPS + lane] = pend1.z; m.ring_[(k + 7) * LS_CLIPS + lane] = pend1.w;
        wr += 32;
        pending = false;
      }
      if (s.st != LS_DONE && wr - s.rd <= LS_RING - 32) {
        pend0 = ls_chunk(base, wr, len2);
        pend1 = ls_chunk(base, wr + 16, len2);
        pending = true;
      }
    }
    if (s.st != LS_DONE && wr - s.rd >= LS_ROUND_BYTES) {
      ls_round(s, m, c, intra_every_round || round % LS_INTRA_PERIOD == 0);
    }
    if (__builtin_amdgcn_ballot_w64(s.st != LS_DONE) == 0) break;
  }
  if (!live) return;
  MobiDevResult r;
  r.rc = 0;
  r.consumed = 0;
  r.pad = 0;
  if (!s.bail) {
    const int used = ls_consumed(s.cbits, len);
    if (used < 0) s.bail = 16;
    else r.consumed = used;
  }
  r.n_intra = s.n_items;
  r.payload_words = s.pay_pos;
  r.quant = s.quant; r.yuvfmt = s.yuvfmt; r.frame_type = (uint32_t)s.iframe;
  if (!s.bail) {
    r.pad = LS_MAGIC;
    MobiDevState *st = A.state_ls + clip; // the shadow copy: mobi_parse_frames moves it into place (unless mobi_ls_deps objects)
    st->quant = s.quant; st->yuvfmt = s.yuvfmt; st->tables_set = s.tables_set; st->frames_started = s.frames_started;
    for (int i = 0; i < 40; i++) st->mcache[i] = m.mc(i);
    st->predx = s.predx; st->predy = s.predy;
    // the MV row cache a P-frame leaves, Internal[221..] -- a later I-frame's walk through Internal[] may read it, mobi_state.h: where the walk
    // kept it (tail_out[clip].mvc) or, from LDS, written there now
    if (!s.iframe && m.mvl) {
      int32_t *mv = A.tail_out[clip].mvc;
      for (int i = 0; i < mvp_words; i++) { const uint32_t v = m.mvl[i * LS_CLIPS + lane]; mv[2 * i] = (int)(int16_t)(v & 0xFFFF); mv[2 * i + 1] = (int)(int16_t)(v >> 16); }
    }
  }
  A.res[clip] = r;
}
// lane = one intra macroblock of a finished clip: workgroup = clip * chunks + chunk
extern "C" __global__ __launch_bounds__(64) void mobi_ls_deps(MobiDevParseArgs A, uint32_t chunks) {
  const uint32_t clip = blockIdx.x / chunks, chunk = blockIdx.x - clip * chunks;
  if (clip >= (uint32_t)A.n_clips || A.bit_len[clip] == MOBI_DP_SKIP) return;
  const MobiDevResult *r = A.res + clip;
  if (r->pad != LS_MAGIC) return;
  const uint32_t idx = chunk * 64 + threadIdx.x, n_mbs = (uint32_t)(A.mbw * A.mbh);
  if (idx >= r->n_intra) return;
  const LsGeom g{A.width, A.height, A.stride, A.lg, A.mbw};
  const uint32_t mb = A.items[(size_t)clip * n_mbs + idx] & 0x1FFFu;
  if (!ls_intra_deps(g, A.desc + (size_t)clip * n_mbs, (int)mb)) A.res[clip].pad = 0; // more than eight: mobi_parse_frames refuses the stream; let it
}

extern "C" int mobi_launch_parse_ls(const MobiDevParseArgs *a, hipStream_t s) {
  if (a->n_clips <= 0) return 0;
  if (a->mbw > 64 || !a->state_ls) return (int)hipErrorInvalidValue;
  // Clips per wave, waves per workgroup.  A wave is one long chain of dependent look-ups: it runs until its slowest lane is done, a round costs
  // what the different states of its lanes need, and alone on a SIMD it issues an instruction every ~20 clocks.  r04 gave every wave a SIMD of
  // its own (32 clips per wave, each with its own 18 KB copy of the tables: 31 ms per P-frame step of 24576 clips, whatever the batch).  r05:
  // the waves of a workgroup share one copy of the tables, and the clips are dealt to about TWO WAVES PER SIMD = 2048 waves -- fewer clips per
  // wave make every wave's life shorter, and two or three such chains interleave on a SIMD for nothing (tools/exp_lsab.sh, 24576 clips: 4 / 6
  // / 8 / 12 / 16 / 32 clips per wave = 40.7 / 43.0 / 27.0 / 25.7 / 27.9 / 31.3 ms -- below 8 the waves no longer all fit the chip at once;
  // 8192 clips: 2 / 4 / 8 / 32 per wave = 26.5 / 18.2 / 29.1 / 31.5 ms; 2048 clips: 1 / 2 = 9.2 / 15.3).
  // A full machine's worth of waves goes out as ONE WORKGROUP OF EIGHT PER CU (256 workgroups).  With two workgroups of four per CU the launch
  // is as fast on its own (25.9 ms either way) but not when the reconstruction of the step before runs beside it (asynchronous steps): a CU
  // that is slow to free LDS and wave slots gets its second workgroup late or not at all, another CU takes a third, and the launch lasts two
  // rounds -- 32768 clips: 68.9 ms per step against 39.9, 24576: 37.4 (16 per wave, r05's first rule) against 35.1, 16384: 34.8 against 28.7,
  // 40960: 55.3 against 47.7 (tools/exp_async.py, profiles/r05_experiments.txt).  49152 clips (24 per wave, what 288 GB hold at 640x480) only
  // fit a CU's LDS this way.
  // LDS per lane without / with the MV row cache's words (r06: the walk keeps the cache's working set in registers and the rest in HBM -- 264 B
  // per lane, full waves of 64 -- or, when the lanes asked for fit with it, in LDS: 432 B at 640 wide, 42 lanes per wave)
  const int per_g = 4 * MOBI_INTRA_RECORDS + LS_RING + 40, per_s = per_g + 4 * (a->mbw + 2);
  const size_t lds_max = 160 * 1024;
  struct Plan { int L, W, turns; double cost; };
  // r06 (frame-parallel groups: n_clips x K virtual clips): more lanes than one such launch holds take TURNS of 256 workgroups, a turn lasts as
  // long as one workgroup lives however few workgroups the last one has, and a wave's life grows slowly with its lanes (640x480 P-frames, ms:
  // 12 lanes 22.9, 24: 27.3, 36: 30.7, 60: 38 -- about 20 + 0.3 per lane): a lane is the cheaper the fuller its wave, and what counts is WHOLE
  // turns -- the fewest that LDS allows, the lanes dealt evenly.  147456 lanes: 36 x 8 in two turns 62 ms; 24 x 8 in three 82; 35 x 8 -- two turns
  // and fifteen workgroups of a third -- 90 (tools/exp_gop_lanes.sh; r06's first sweep read that as "beyond 24 lanes a wave costs what its lanes
  // bring": it had measured the tail).  More than two waves per SIMD lose what they gain (15 x 8 twice per CU: 107: the walk is issue-bound at two).
  auto plan = [&](int per_lane, double penalty) {
    auto lds_of = [&](int w, int l) { return (size_t)MOBI_DT_BYTES + (size_t)w * l * per_lane; };
    Plan p;
    p.L = (a->n_clips + 2047) / 2048;
    p.L = p.L < 1 ? 1 : p.L > 64 ? 64 : p.L;
    p.W = MOBI_LS_WAVES;
    p.turns = 1;
    if ((a->n_clips + p.L - 1) / p.L > 1536 && lds_of(2 * p.W, p.L) <= lds_max) p.W *= 2;
    if (lds_of(p.W, p.L) > lds_max || p.L > 24) {
      p.W = 2 * MOBI_LS_WAVES;
      int Lmax = 64;
      while (Lmax > 1 && lds_of(p.W, Lmax) > lds_max) Lmax--;
      const long per_turn = 256L * p.W; // waves of one turn (one workgroup per CU)
      p.turns = (int)((a->n_clips + Lmax * per_turn - 1) / (Lmax * per_turn));
      p.L = (int)((a->n_clips + p.turns * per_turn - 1) / (p.turns * per_turn));
    }
    p.cost = p.turns * (20.0 + 0.3 * p.L) * penalty;
    return p;
  };
  // (the round trip to HBM for one word per macroblock costs a wave ~5 % of its life: 49152 lanes 29.4 against 28.0 ms, 73728: 32.3 against 30.8)
  const Plan pg = plan(per_g, 1.05), ps = plan(per_s, 1.0);
  int mv_lds = ps.cost <= pg.cost;
  int L = mv_lds ? ps.L : pg.L, W = mv_lds ? ps.W : pg.W;
#if defined(MOBI_PROFILING)
  if (const char *e = getenv("MOBI_LS_CLIPS")) L = atoi(e);    // (tools/exp_lsab.sh, tests/test_lsparse_gpu.py)
  if (const char *e = getenv("MOBI_LS_WG_WAVES")) W = atoi(e);
#endif
  if (L < 1 || L > 64 || W < 1 || W > 8) return (int)hipErrorInvalidValue;
  if ((size_t)MOBI_DT_BYTES + (size_t)W * L * per_s > lds_max) mv_lds = 0; // (a forced shape that only fits without the cache)
  MobiDevParseArgs b = *a;
  b.ls_clips = L;
  b.ls_mv_lds = mv_lds;
  const size_t lds = (size_t)MOBI_DT_BYTES + (size_t)W * L * (mv_lds ? per_s : per_g);
  if (lds > lds_max) return (int)hipErrorInvalidValue;
  if (lds > 64 * 1024) // (per device and per size; cheap)
    if (hipFuncSetAttribute((const void *)mobi_parse_frames_ls, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return (int)hipGetLastError();
  const dim3 grid((unsigned)((a->n_clips + L * W - 1) / (L * W)));
  hipLaunchKernelGGL(mobi_parse_frames_ls, grid, dim3(64 * W), lds, s, b);
  const uint32_t chunks = (uint32_t)(a->mbw * a->mbh + 63) / 64;
  hipLaunchKernelGGL(mobi_ls_deps, dim3((unsigned)a->n_clips * chunks), dim3(64), 0, s, *a, chunks);
  return (int)hipGetLastError();
}
