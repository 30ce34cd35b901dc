// mobi_abi.cpp -- C ABI of libmobiclip_hip.so (include/mobiclip_hip.h): owns the HIP buffers, runs the
// host bitstream parse and launches the reconstruction kernels.  There is no CPU reconstruction path:
// every entry point that produces pixels goes through mobi_launch_inter / mobi_launch_intra.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/mobiclip_hip.h"
#include "mobi_cmd.h"
#include "mobi_dparse.h"
#define MOBI_GOP_DEVICE_DECLS
#include "mobi_gop.h"
#include "mobi_kernels.h"
#include "mobi_tile.h"
#include "mobi_parse.h"

namespace {

constexpr size_t kPaySlack = 2048; // lanes of the inter kernel with nothing to fetch re-read up to ~1 KB past their macroblock's payload
constexpr size_t kGuard = 65536; // slack on both ends of the plane arena: the slow MC path fetches the dwords of the row below a window's last
                                 // row whether its phase needs them or not, and in the tiled planes (mobi_tile.h) that row may be a tile row (<= 16 KB) away
constexpr size_t kAlign = 16;
constexpr size_t kFusedStepMbs = (size_t)256 * 1200; // steps of at most this many macroblocks go out as ONE launch (mobi_recon_step); MOBI_FUSED_STEP_MBS overrides, 0 = never

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

#define HIP_TRY(expr)                                                             \
  do {                                                                            \
    hipError_t e_ = (expr);                                                       \
    if (e_ != hipSuccess) {                                                       \
      snprintf(g_last_hip_error, sizeof(g_last_hip_error), "%s: %s", #expr, hipGetErrorString(e_)); \
      (void)hipGetLastError(); /* HIP keeps the error until somebody reads it: the next launch wrapper's hipGetLastError() must not find this one \
                                  (r05: an allocation that failed in one batch made the first launch of the NEXT batch report "out of memory") */ \
      return MOBI_E_DEVICE;                                                       \
    }                                                                             \
  } while (0)
thread_local char g_last_hip_error[256] = "";

struct PinnedBuf { // growable pinned host staging buffer
  uint8_t *p = nullptr;
  size_t cap = 0;
  int reserve(size_t n) {
    if (n <= cap) return MOBI_OK;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    size_t want = std::max(n, (size_t)1 << 20);
    HIP_TRY(hipHostMalloc((void **)&p, want, hipHostMallocDefault));
    cap = want;
    return MOBI_OK;
  }
  ~PinnedBuf() { if (p) (void)hipHostFree(p); }
};
struct DevBuf {
  uint8_t *p = nullptr;
  size_t cap = 0;
  int reserve(size_t n) {
    if (n <= cap) return MOBI_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = std::max(n, (size_t)1 << 20);
    HIP_TRY(hipMalloc((void **)&p, want));
    cap = want;
    return MOBI_OK;
  }
  ~DevBuf() { if (p) (void)hipFree(p); }
};

// One frame step on the device = [MbDesc table: n_clips * n_mbs][payload arena: all clips back to back].
// Clips whose parse failed (nullptr) get descriptors typed "intra" that no launch list references, so
// nothing of theirs is ever written (the reference leaves a partial frame; content is unspecified here).
size_t step_payload_words(const std::vector<const ParsedFrame *> &frames) {
  size_t w = 0;
  for (auto *f : frames)
    if (f) w += f->payload.size();
  return w;
}
// one clip's share of the step image; base = where its payload starts in the arena
void step_write_clip(const ParsedFrame *f, size_t base, int n_mbs, MbDesc *dd, uint32_t *payload) {
  if (!f) {
    for (int i = 0; i < n_mbs; i++) dd[i] = MbDesc{0, MOBI_MB_INTRA, 0, 0, 0, 0, 0, 0};
    return;
  }
  for (int i = 0; i < n_mbs; i++) {
    dd[i] = f->desc[i];
    dd[i].payload_off += (uint32_t)base;
  }
  if (!f->payload.empty()) memcpy(payload + base, f->payload.data(), f->payload.size() * 4);
}
void step_write(const std::vector<const ParsedFrame *> &frames, int n_mbs, MbDesc *desc, uint32_t *payload) {
  size_t base = 0;
  for (size_t c = 0; c < frames.size(); c++) {
    step_write_clip(frames[c], base, n_mbs, desc + c * (size_t)n_mbs, payload);
    if (frames[c]) base += frames[c]->payload.size();
  }
}

struct LevelPlan { // launch plan of one frame step: the intra macroblocks of all clips, sorted by dependency level
  // One item = MOBI_INTRA_ITEM_WORDS words, everything a wave needs to start (mobi_recon_intra): (clip << 13 | mb), MbDesc.w1,
  // MbDesc.payload_off inside the step's arena, and flags: [0] 16x16 plane, [1] has intra dependencies (must poll their tags),
  // [2] has intra dependents (must publish its own), [14:5] number of level words,
  // [31:16] plane parameter.
  std::vector<uint32_t> items;
  uint32_t n_items = 0;         // launch items, padding included
  uint32_t n_intra = 0;         // intra macroblocks among them
  bool any_inter = false;
  uint64_t cmd_bytes = 0;       // descriptors + payload of every macroblock of the step
  uint64_t intra_cmd_bytes = 0; // ... of the intra ones: descriptor, 24 block records, level words (the inter kernel never reads them)
  void build(const std::vector<const ParsedFrame *> &frames, int /*mbw*/) {
    // Few intra macroblocks (small batches): every one gets a wave of its own (three null items behind it) -- the chip has room for
    // them all at once, and a lone macroblock's wave is shorter than a wave that runs the longest step list of four.  Measured
    // (640x480 P-frames, intra launch): 8 clips 45 -> 33 us, 64 clips 53 -> 41 us, 512 clips 78 -> 110 us: dense from there on.
#if defined(MOBI_PROFILING)
    static const int sparse_env = getenv("MOBI_INTRA_SPARSE") ? atoi(getenv("MOBI_INTRA_SPARSE")) : -1; // (measurements only)
#else
    const int sparse_env = -1;
#endif
    uint64_t total_intra = 0;
    for (auto *f : frames)
      if (f) total_intra += f->hdr.n_intra;
    const bool sparse = sparse_env >= 0 ? sparse_env != 0 : total_intra <= 4096;
    uint32_t maxl = 0;
    any_inter = false;
    cmd_bytes = 0;
    intra_cmd_bytes = 0;
    for (auto *f : frames)
      if (f) {
        maxl = std::max(maxl, f->hdr.n_levels);
        if (f->hdr.n_intra < f->hdr.n_mbs) any_inter = true;
        cmd_bytes += f->hdr.cmd_bytes;
      }
    items.clear();
    std::vector<size_t> base(frames.size() + 1, 0); // where each clip's payload starts in the step's arena (as step_write lays it out)
    for (size_t c = 0; c < frames.size(); c++) base[c + 1] = base[c] + (frames[c] ? frames[c]->payload.size() : 0);
    n_intra = 0;
    size_t total = 0;
    for (auto *f : frames)
      if (f) total += f->intra_items.size();
    items.reserve((sparse ? 4 : 1) * total + 16 * (size_t)maxl);
    const uint32_t none[4] = {MOBI_ITEM_NONE, 0, 0, 0};
    for (uint32_t L = 1; L <= maxl; L++) {
      // inside a level by class (mobi_parse.cpp, finish_levels): macroblocks away from the picture's edges first, those nobody depends on first, by the number of split areas
      // (a wave runs as many steps as the longest of its four macroblocks has, and publishes if any of them must); the ones at an edge (on Width == Stride pictures their halo
      // needs the per-sample ownership test, mobi_recon_intra) behind them, so that few waves of four carry one.  The parsers wrote the items
      // (ParsedFrame::intra_items): this is a concatenation with the clip number and the clip's place in the arena added.
      for (uint32_t k = 0; k < MOBI_INTRA_CLASSES; k++)
        for (size_t c = 0; c < frames.size(); c++) {
          const ParsedFrame *f = frames[c];
          if (!f || L > f->hdr.n_levels) continue;
          for (uint32_t i = f->class_start[(size_t)L * MOBI_INTRA_CLASSES + k]; i < f->class_start[(size_t)L * MOBI_INTRA_CLASSES + k + 1]; i++) {
            const uint32_t *it = &f->intra_items[(size_t)i * 4];
            intra_cmd_bytes += sizeof(MbDesc) + 4 * (MOBI_INTRA_RECORDS + ((it[3] >> 5) & 0x3FFu));
            const uint32_t item[4] = {MOBI_ITEM(c, it[0]), it[1], it[2] + (uint32_t)base[c], it[3] & ~8u};
            items.insert(items.end(), item, item + 4);
            n_intra++;
            for (int q = 1; q < 4 && sparse; q++) items.insert(items.end(), none, none + 4);
          }
        }
      // a wave carries four macroblocks, and a macroblock may wait for one of the level before: levels start on a wave boundary
      while ((items.size() / MOBI_INTRA_ITEM_WORDS) & 3) items.insert(items.end(), none, none + 4);
    }
    n_items = (uint32_t)(items.size() / MOBI_INTRA_ITEM_WORDS);
  }
};

} // namespace

// Host-side parse pool: the VLC parse is serial inside a clip but clips are independent (MD.cs:15-39), so the clips of a
// batch are parsed by a few persistent threads (the reference runs one decode thread per open file, Form1.cs:199-215).
class ParsePool {
 public:
  explicit ParsePool(int n_threads) {
    for (int i = 0; i < n_threads; i++) th_.emplace_back([this] { worker(); });
  }
  ~ParsePool() {
    { std::lock_guard<std::mutex> l(m_); stop_ = true; gen_++; }
    cv_.notify_all();
    for (auto &t : th_) t.join();
  }
  int size() const { return (int)th_.size(); }
  // runs f(0..n-1), the caller's thread included; returns when all are done
  void run(int n, const std::function<void(int)> &f) {
    if (th_.empty() || n < 2) { for (int i = 0; i < n; i++) f(i); return; }
    { std::lock_guard<std::mutex> l(m_); job_ = &f; n_ = n; next_.store(0); busy_ = (int)th_.size(); gen_++; }
    cv_.notify_all();
    drain();
    std::unique_lock<std::mutex> l(m_);
    done_.wait(l, [this] { return busy_ == 0; });
    job_ = nullptr;
  }

 private:
  void drain() {
    for (int i; (i = next_.fetch_add(1)) < n_;) (*job_)(i);
  }
  void worker() {
    uint64_t seen = 0;
    for (;;) {
      { std::unique_lock<std::mutex> l(m_); cv_.wait(l, [&] { return gen_ != seen; }); seen = gen_; if (stop_) return; }
      drain();
      { std::lock_guard<std::mutex> l(m_); if (--busy_ == 0) done_.notify_one(); }
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  const std::function<void(int)> *job_ = nullptr;
  std::atomic<int> next_{0};
  int n_ = 0, busy_ = 0;
  uint64_t gen_ = 0;
  bool stop_ = false;
};

struct mobi_batch {
  int n = 0, device = 0, version = 0;
  MobiGeom g{};
  hipStream_t stream = nullptr;
  uint8_t *arena = nullptr; // guard | clips x 6 slots | guard
  size_t slot_bytes = 0, clip_bytes = 0;
  int ring_base = 0;
  int frames_started = 0;
  bool poisoned = false; // an asynchronous submit failed half-way (mobi_batch_submit): nothing further is accepted
  int debug = 0;
  std::vector<std::unique_ptr<MobiStreamParser>> parsers;
  std::vector<ParsedFrame> cur; // per clip, current frame (batch_decode)
  int *d_fault = nullptr;
  std::vector<int> h_fault;
  // per-call staging (batch_decode)
  PinnedBuf h_stage;
  PinnedBuf h_items;                   // host-parsed steps: the launch list (the staging buffer's chunks are in flight by then)
  DevBuf d_cmd, d_items;
  int32_t *d_scale = nullptr; // [MOBI_SCALE_QMAX][MOBI_SCALE_STRIDE]
  unsigned long long *d_prof = nullptr; // MOBI_DEBUG=9: in-kernel cycle accumulators
  std::unique_ptr<ParsePool> pool;      // host parse threads (MOBI_PARSE_THREADS, default min(clips, cores, 32) - 1 helpers)
  uint32_t *d_argb = nullptr;           // Bitmap output of mobi_batch_convert_argb / mobi_batch_get_argb (lazily allocated)
  size_t argb_bytes = 0;
  bool argb_all_valid = false;          // d_argb holds every clip's Bitmap of the current frame
  uint8_t *d_lin = nullptr;             // one frame in the reference's linear layout: what mobi_batch_get_planes copies out (lazily allocated)
  uint32_t *d_done = nullptr;           // [clip * n_mbs + mb]: step tag of the last step that reconstructed this intra macroblock
  uint32_t step_tag = 0;                // bumped once per frame step, never 0
  // device-side parse (mobi_dparse.hip): parse_mode 1 = mobi_batch_decode parses on the GPU (env MOBI_DEVICE_PARSE=1 or
  // mobi_batch_set_parse_mode); the per-clip decoder state then lives in d_pstate and the host parsers stay untouched
  int parse_mode = 0; // explicit (env / mobi_batch_set_parse_mode), or by batch size and settled at the first frame (parse_auto)
  bool parse_auto = false;
  int hybrid_host = 0;           // parse mode 2: the last hybrid_host clips are parsed by the host pool while the GPU parses the others
  hipStream_t stream2 = nullptr; // their command lists go up on this stream, beside the parse kernel
  hipEvent_t ev_up = nullptr;
  PinnedBuf h_stage2;
  size_t last_pay_cap = 0;
  float last_parse_ms = 0; // duration of the last mobi_parse_frames launch (kernel timing on)
  float last_decode_ms = 0, last_stage_ms = 0; // wall time of the last mobi_batch_decode call / of its host staging part
  float phase_ms[6] = {0, 0, 0, 0, 0, 0};      // (profiling) host-parsed step: parse loop, plan, stage + upload enqueue, launch enqueue, sync; device-parsed step: cumulative ms after gather + upload, parse enqueued, parse done, launches enqueued, all done
  float last_hostparse_ms = 0;                 // ... / of its host parse part (host parse mode)
  DevBuf d_bits, d_pdesc, d_ppay, d_pitems;
  DevBuf d_src, d_search; // mobi_batch_motion_search: the pictures being analysed, the packed results
  // the decoder state of the device-parsed clips: a ring of three entries (a step reads ps_cur, writes ps_cur + 1), so that the state a frame
  // STARTED from is still there when the host parser has to take the frame over (decode_device_parse, async_repair; mobi_state.h)
  MobiDevState *d_pstate[3] = {nullptr, nullptr, nullptr};
  MobiDevTail *d_ptail[3] = {nullptr, nullptr, nullptr};
  int ps_cur = 0;
  MobiDevState *d_pstate_ls = nullptr; // shadow copy the lock-step parser writes (mobi_lsparse.hip)
  std::vector<uint8_t> on_host;        // [clip] device-parse modes: this clip is the host parser's (the hybrid mode's share; clips with a frame the
                                       // device parser could not finish, from that frame on)
  unsigned long fallbacks = 0;         // frames taken over so far
  unsigned long returns = 0;           // clips handed back to the device parsers so far (dp_return)
  std::vector<uint8_t> host_share;     // [clip] 1: the hybrid mode's share (stays with the host parser by design)
  std::vector<uint16_t> clean_run, clean_need; // [clip] consecutive frames the device parsers could have finished; how many it takes to go back
  PinnedBuf h_ret;                     // dp_return: states on their way back to the device
  PinnedBuf h_fix;                     // async_repair: one clip's command list
  DevBuf d_fix;
  uint32_t pay_clip_words = 0;         // of the last device-parsed step (MobiReconArgs.pay_clip_words)
  int ls_finished = -1;                // clips of the last step it finished itself
  int ls_policy = 0;                   // 0: never (parse modes 1, 2 named by the caller); 1: always (mode 3); 2: by step (the default, ls_decide)
  bool lockstep = false;               // this step: mobi_parse_frames_ls in front of mobi_parse_frames
  int host_chunk = 256;                // host-parsed steps: clips per chunk of the parse / stage / upload pipeline (mobi_batch_decode)
  size_t fused_mbs = kFusedStepMbs;     // launch_plan: steps of at most this many macroblocks go out as one launch
  MobiDevResult *d_pres = nullptr;
  uint8_t *d_ptables = nullptr;
  PinnedBuf h_pres;
  std::vector<uint32_t> dev_quant, dev_yuvfmt;
  // asynchronous steps (mobi_batch_submit / mobi_batch_wait, device parse only): two sets of staging so that the bytes of step
  // n + 1 are gathered and uploaded while the GPU works on step n; everything else follows stream order
  struct AsyncSlot {
    PinnedBuf h_stage, h_pres, h_fault, h_over, h_ret; // h_over: the command lists of the host parser's clips (dp_override); h_ret: dp_return
    DevBuf d_bits;
    // what the parse of this step leaves in HBM: owned by the slot, so that the parse of step n + 1 (on stream_p) may run under the
    // reconstruction of step n (on stream), which still reads step n's
    DevBuf d_pdesc, d_ppay, d_pitems, d_pres;
    hipEvent_t ev_up = nullptr, ev_done = nullptr, ev_parsed = nullptr;
    std::vector<int32_t> offs; // Offset of every clip at submission
    int n_dev = 0;
    bool lockstep = false;           // the lock-step parser was in front of THIS step (ls_decide at submission)
    bool parsed_recorded = false;    // ev_parsed has been recorded at least once
    int state_in = 0, ring_base = 0; // the entry of the state ring this step's parse read; the ring position its reconstruction wrote
    size_t hdr_bytes = 0;            // of the staged bitstream image (h_stage: offsets, lengths, bits)
    // clips whose result is the host parser's (its own clips at submission; clips repaired in mobi_batch_wait)
    std::vector<uint8_t> is_host;
    std::vector<int> host_rc;
    std::vector<int32_t> host_off;
    std::vector<uint32_t> host_quant, host_yuv; // Quantizer / YuvFormat behind that frame (the parser itself may be a step further by the time of wait)
  };
  AsyncSlot aslot[2];
  // frame-parallel groups (mobi_batch_gop_begin / mobi_batch_gop_finish, mobi_gop.h): K frames of every clip parsed side by side as n * K
  // virtual clips (v = k * n + c), reconstructed as K steps.  Two slots: the bytes of group g + 1 are gathered and uploaded, and its parse
  // enqueued, while group g is reconstructed.
  struct GopSlot {
    PinnedBuf h_stage, h_res, h_fault, h_over[MOBI_GOP_PARSE_MAX], h_seed, h_ret;
    DevBuf d_bits, d_desc, d_pay, d_items, d_res, d_sin, d_sout, d_sls, d_tails, d_fault;
    hipEvent_t ev_up = nullptr, ev_parsed = nullptr, ev_recon = nullptr;
    int K = 0;
    size_t hdr_bytes = 0, bytes = 0, max_len = 0, cap_words = 0;
    int n_iframes = 0;
    bool parse_enqueued = false, lockstep = false;
    int ring_in = 0;                 // the state ring entry the group's parse read
    std::vector<uint64_t> boff;      // [v] where the frame's bytes start in the staged image (behind its header)
    std::vector<uint32_t> lens;      // [v] their number (the header's copy carries MOBI_DP_SKIP for the host parser's clips)
    std::vector<int32_t> offs;       // [v] Offset at submission
    std::vector<uint8_t> is_host;    // [c] the host parser's clip when the parse was enqueued
    // what the group's first mobi_batch_gop_finish settles for all K frames (a group of more than six is finished in two calls)
    bool resolved = false, returned = false; // (returned: the clips that go back to the device parsers have been sent)
    bool sorted = false;             // the group's intra macroblocks were put in wavefront order on the device (mobi_launch_gop_sort)
    DevBuf d_sorted, d_hist;
    uint64_t sorted_off[MOBI_GOP_PARSE_MAX] = {0};
    uint32_t sorted_items[MOBI_GOP_PARSE_MAX] = {0};
    int done = 0;                    // frames reconstructed and reported so far
    std::vector<int> host_from, hslot, hrc, all_host;
    std::vector<int32_t> hoff;
    std::vector<uint32_t> hq, hy;
    std::vector<uint8_t> hready;
  };
  GopSlot gslot[2];
  int gop_head = 0, gop_count = 0;
  std::vector<ParsedFrame> gop_frames; // host-parsed frames of the group being finished
  hipStream_t stream_p = nullptr;      // asynchronous steps: the parse kernels (upload on stream2, reconstruction on stream)
  int async_head = 0, async_count = 0; // oldest step in flight, number of steps in flight (<= 2)
  size_t dp_len_hint = 0;              // longest frame seen so far (+ 25 %): sizes the payload arena of the device-side parser
  uint64_t async_seq = 0;
  // preloaded replay
  // [clip] -> frames; clones share the host copy (each clip still gets its own bytes in HBM at commit)
  std::vector<std::shared_ptr<std::vector<ParsedFrame>>> staged;
  std::vector<std::shared_ptr<std::vector<int>>> staged_rc;
  int n_frames_loaded = 0;
  DevBuf r_cmd, r_items;
  std::vector<LevelPlan> r_plan;        // per frame
  std::vector<size_t> r_items_off;      // per frame: word offset of its items in r_items
  std::vector<size_t> r_desc_off, r_payload_off; // per frame: byte offsets inside r_cmd
  bool committed = false;
  // timing
  hipEvent_t ev_begin = nullptr, ev_end = nullptr;
  hipEvent_t ev_p0 = nullptr, ev_p1 = nullptr; // around the parse launch (kernel timing on)
  int ktiming = 0; // HIP events around launches: 0 none, 1 the inter launches (the dominant kernel: roofline), 2 every launch
  struct EvPair { hipEvent_t a, b; int kind; };
  std::vector<EvPair> evs;
  std::vector<hipEvent_t> ev_pool;
  float acc_ms[2] = {0, 0};
  int acc_launches[2] = {0, 0};

  MobiReconArgs args(const uint8_t *desc, const uint8_t *payload) const {
    MobiReconArgs a;
    memset(&a, 0, sizeof(a));
    a.planes = arena + kGuard;
    a.desc = (const MbDesc *)desc;
    a.payload = (const uint32_t *)payload;
    a.scale = d_scale;
    a.fault = d_fault;
    a.clip_bytes = clip_bytes;
    a.slot_bytes = (uint32_t)slot_bytes;
    a.ring_base = ring_base;
    a.width = g.width; a.height = g.height; a.stride = g.stride; a.mbw = g.mbw;
    a.n_mbs = g.mbw * g.mbh;
    a.n_clips = n;
    auto magic = [](uint32_t d) { uint64_t m = ((uint64_t)1 << 32) / d; return (uint32_t)(m > 0xFFFFFFFFull ? 0xFFFFFFFFull : m); }; // d == 1 must not wrap to 0
    a.step_tag = step_tag;
    a.done = d_done;
    a.prof = d_prof;
    a.qpr = (uint32_t)(a.mbw + 3) / 4;
    a.qpc = a.qpr * (uint32_t)g.mbh;
    a.magic_qpr = magic(a.qpr);
    a.magic_qpc = magic(a.qpc);
    return a;
  }
  hipEvent_t get_event() {
    if (!ev_pool.empty()) { hipEvent_t e = ev_pool.back(); ev_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
  }
  // one frame step = the inter launch, then ONE intra launch for all dependency levels (items sorted by level, waves wait for
  // the tags of the macroblocks they depend on).  r01 also had a launch per level and a whole-step launch; both were slower.
  bool last_fused = false; // the last launch_plan went out as one launch (mobi_recon_step)
  int launch_plan(const MobiReconArgs &a, const LevelPlan &plan, const uint32_t *items_dev, int n_clips = -1, bool allow_fused = true) {
    if (n_clips < 0) n_clips = n;
    last_fused = false;
    // Small steps (BASELINE config 4: 8 clips per GPU): one launch carries both kinds of macroblock -- the kernel boundary between the two
    // launches is a fifth of such a step.  Large ones keep two: the fused kernel has the octet's registers and LDS for the intra fours too.
    if (allow_fused && plan.any_inter && plan.n_items && (size_t)n_clips * g.mbw * g.mbh <= fused_mbs) {
      last_fused = true;
      EvPair ep{nullptr, nullptr, 0};
      if (ktiming) { ep.a = get_event(); ep.b = get_event(); (void)hipEventRecord(ep.a, stream); }
      if (mobi_launch_step(&a, items_dev, (int)plan.n_items, stream) != 0) return MOBI_E_DEVICE;
      if (ktiming) { (void)hipEventRecord(ep.b, stream); evs.push_back(ep); }
      return MOBI_OK;
    }
    if (plan.any_inter) {
      EvPair ep{nullptr, nullptr, 0};
      if (ktiming) { ep.a = get_event(); ep.b = get_event(); (void)hipEventRecord(ep.a, stream); }
      if (mobi_launch_inter(&a, stream) != 0) return MOBI_E_DEVICE;
      if (ktiming) { (void)hipEventRecord(ep.b, stream); evs.push_back(ep); }
    }
    if (plan.n_items) {
      EvPair ep{nullptr, nullptr, 1};
      if (ktiming >= 2) { ep.a = get_event(); ep.b = get_event(); (void)hipEventRecord(ep.a, stream); }
      if (mobi_launch_intra(&a, items_dev, (int)plan.n_items, stream) != 0) return MOBI_E_DEVICE;
      if (ktiming >= 2) { (void)hipEventRecord(ep.b, stream); evs.push_back(ep); }
    }
    return MOBI_OK;
  }
  void drain_events() {
    for (auto &e : evs) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) { acc_ms[e.kind] += ms; acc_launches[e.kind]++; }
      ev_pool.push_back(e.a);
      ev_pool.push_back(e.b);
    }
    evs.clear();
  }
  ~mobi_batch() {
    if (stream2) (void)hipStreamSynchronize(stream2); // an upload of a submitted step may still be reading pinned memory we are about to free
    if (stream_p) (void)hipStreamSynchronize(stream_p);
    if (stream) (void)hipStreamSynchronize(stream);
    drain_events();
    for (auto e : ev_pool) (void)hipEventDestroy(e);
    if (ev_up) (void)hipEventDestroy(ev_up);
    for (auto &gs : gslot) {
      if (gs.ev_up) (void)hipEventDestroy(gs.ev_up);
      if (gs.ev_parsed) (void)hipEventDestroy(gs.ev_parsed);
      if (gs.ev_recon) (void)hipEventDestroy(gs.ev_recon);
    }
    for (auto &sl : aslot) {
      if (sl.ev_up) (void)hipEventDestroy(sl.ev_up);
      if (sl.ev_done) (void)hipEventDestroy(sl.ev_done);
      if (sl.ev_parsed) (void)hipEventDestroy(sl.ev_parsed);
    }
    if (stream2) (void)hipStreamDestroy(stream2);
    if (stream_p) (void)hipStreamDestroy(stream_p);
    if (ev_p0) (void)hipEventDestroy(ev_p0);
    if (ev_p1) (void)hipEventDestroy(ev_p1);
    if (ev_begin) (void)hipEventDestroy(ev_begin);
    if (ev_end) (void)hipEventDestroy(ev_end);
    if (arena) (void)hipFree(arena);
    if (d_fault) (void)hipFree(d_fault);
    if (d_scale) (void)hipFree(d_scale);
    if (d_prof) (void)hipFree(d_prof);
    if (d_done) (void)hipFree(d_done);
    if (d_lin) (void)hipFree(d_lin);
    if (d_argb) (void)hipFree(d_argb);
    for (int k = 0; k < 3; k++) {
      if (d_pstate[k]) (void)hipFree(d_pstate[k]);
      if (d_ptail[k]) (void)hipFree(d_ptail[k]);
    }
    if (d_pstate_ls) (void)hipFree(d_pstate_ls);
    if (d_pres) (void)hipFree(d_pres);
    if (d_ptables) (void)hipFree(d_ptables);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

struct mobi_dec { mobi_batch *b; };

extern "C" {

// ---- test and profiling hooks: only in the -DMOBI_PROFILING twin of the library (libmobiclip_hip_prof.so); the product library has none
// of them (mobi_debug_write_planes lets any caller overwrite reference frames) ----
#if defined(MOBI_PROFILING)
#pragma GCC visibility push(default)
// profiling aid, not part of the public header: copy out the MOBI_DEBUG=9 per-wave cycle records
// (uint32 x 4 per macroblock: descriptor, pixels+MC, residual, store drain) of the last inter launch
int mobi_debug_read_prof(mobi_batch *b, uint32_t *out, size_t n_words) {
  if (!b || !b->d_prof) return MOBI_E_ARG;
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipStreamSynchronize(b->stream));
  HIP_TRY(hipMemcpy(out, b->d_prof, n_words * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemset(b->d_prof, 0, n_words * 4)); // read and clear: the next read sees only the steps in between
  return MOBI_OK;
}

// test aid, not part of the public header: overwrite ring slot `ring_idx` of one clip with planes given in the reference layout
// (Y: Stride * Height, UV: Stride * Height / 2) -- tests/test_unit_vectors.py predicts intra macroblocks from chosen halos this way.
// The caller keeps the padding zero (MD.cs:107: fresh planes are zeroed and nothing ever writes the padding).
int mobi_debug_write_planes(mobi_batch *b, int clip, int ring_idx, const uint8_t *y, const uint8_t *uv) {
  if (!b || clip < 0 || clip >= b->n || ring_idx < 0 || ring_idx > 5 || !y || !uv) return MOBI_E_ARG;
  if (ring_idx >= b->frames_started) return MOBI_E_NULLREF;
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipStreamSynchronize(b->stream));
  const uint32_t ysz = (uint32_t)b->g.stride * (uint32_t)b->g.height;
  std::vector<uint8_t> t(b->slot_bytes);
  for (uint32_t a = 0; a < ysz; a++) t[mobi_ty(a, b->g.lg)] = y[a];
  for (uint32_t a = 0; a < ysz / 2; a++) t[ysz + mobi_tc(a, b->g.lg)] = uv[a];
  uint8_t *slot = b->arena + kGuard + (size_t)clip * b->clip_bytes + (size_t)((b->ring_base + 6 - ring_idx) % 6) * b->slot_bytes;
  HIP_TRY(hipMemcpy(slot, t.data(), t.size(), hipMemcpyHostToDevice));
  b->argb_all_valid = false;
  return MOBI_OK;
}
mobi_batch *mobi_debug_dec_batch(mobi_dec *d) { return d ? d->b : nullptr; }
// test aid, not part of the public header: what the last device-side parse left in HBM.  desc_out: n_clips*n_mbs*8 words,
// items_out: n_clips*n_mbs words, res_out: n_clips*8 words, payload_out: n_clips*pay_cap words (any may be null); returns pay_cap
float mobi_debug_parse_ms(const mobi_batch *b) { return b ? b->last_parse_ms : 0.f; }
float mobi_debug_stage_ms(const mobi_batch *b) { return b ? b->last_stage_ms : 0.f; }
float mobi_debug_hostparse_ms(const mobi_batch *b) { return b ? b->last_hostparse_ms : 0.f; }
float mobi_debug_phase_ms(const mobi_batch *b, int k) { return b && k >= 0 && k < 6 ? b->phase_ms[k] : 0.f; } // host-parsed step: where the call's time went
long long mobi_debug_read_parse(mobi_batch *b, uint32_t *desc_out, uint32_t *items_out, uint32_t *res_out, uint32_t *payload_out, size_t payload_words) {
  if (!b || !b->d_pres) return MOBI_E_ARG;
  const size_t n = (size_t)b->n, n_mbs = (size_t)b->g.mbw * b->g.mbh;
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipStreamSynchronize(b->stream));
  if (desc_out) HIP_TRY(hipMemcpy(desc_out, b->d_pdesc.p, n * n_mbs * sizeof(MbDesc), hipMemcpyDeviceToHost));
  if (items_out) HIP_TRY(hipMemcpy(items_out, b->d_pitems.p, n * n_mbs * 4, hipMemcpyDeviceToHost));
  if (res_out) HIP_TRY(hipMemcpy(res_out, b->d_pres, n * sizeof(MobiDevResult), hipMemcpyDeviceToHost));
  if (payload_out) HIP_TRY(hipMemcpy(payload_out, b->d_ppay.p, std::min(payload_words * 4, b->d_ppay.cap), hipMemcpyDeviceToHost));
  return (long long)b->last_pay_cap;
}
#pragma GCC visibility pop
#endif // MOBI_PROFILING

const char *mobi_build_info(void) { return "libmobiclip_hip 0.6 (gfx950, macroblock-tiled planes; HIP kernels: mobi_recon_inter8, mobi_recon_intra, mobi_recon_step, mobi_recon_intra_cl, mobi_recon_intra_walk, mobi_parse_frames, mobi_parse_frames_ls, mobi_ls_deps, mobi_parse_tail, mobi_gop_prepare, mobi_gop_chain, mobi_gop_fronts, mobi_gop_front_starts, mobi_gop_scatter, mobi_untile, mobi_yuv_to_argb, mobi_motion_search_2x2, mobi_fwd_dct8, mobi_fwd_dct4; no CPU reconstruction path)"; }

const char *mobi_error_string(int rc) {
  switch (rc) {
    case MOBI_OK: return "ok";
    case MOBI_E_INDEX: return "index out of range (reference would throw IndexOutOfRange/ArgumentException)";
    case MOBI_E_NULLREF: return "reference frame slot is null";
    case MOBI_E_PARTCODE: return "illegal partition code";
    case MOBI_E_VERSION: return "unsupported codec version";
    case MOBI_E_CLAMP: return "residual left the clamp-table domain";
    case MOBI_E_UNSUPPORTED: return "a walk through Internal[] left a coefficient beyond int16 that need not leave the clamp table's domain (the one input outside the parity domain)";
    case MOBI_E_ARG: return "bad argument";
    case MOBI_E_DEVICE: return g_last_hip_error[0] ? g_last_hip_error : "HIP error";
    default: return "unknown";
  }
}

mobi_batch *mobi_batch_create(int n_clips, uint32_t width, uint32_t height, int version, int device) {
  if (n_clips < 1 || n_clips >= (1 << 19) || width == 0 || height == 0 || (width & 15) || (height & 15) || width > 1024 ||
      (width / 16) * (height / 16) > 8191)
    return nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
    snprintf(g_last_hip_error, sizeof(g_last_hip_error), "no usable HIP device (count=%d, requested %d)", ndev, device);
    return nullptr;
  }
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  auto b = std::make_unique<mobi_batch>();
  b->n = n_clips;
#if defined(MOBI_PROFILING)
  if (const char *dbg = getenv("MOBI_DEBUG")) b->debug = atoi(dbg); // 9: the octet kernel with in-kernel cycle records (tools/exp_prof8.py)
#endif
  b->device = device;
  b->version = version;
  for (int i = 0; i < n_clips; i++) b->parsers.emplace_back(new MobiStreamParser(width, height, version));
  b->g = b->parsers[0]->geom();
  b->slot_bytes = (size_t)b->g.stride * b->g.height * 3 / 2;
  b->clip_bytes = 6 * b->slot_bytes;
  b->cur.resize(n_clips);
  b->h_fault.assign(n_clips, 0);
  b->on_host.assign(n_clips, 0);
  size_t arena_clips = (size_t)n_clips;
  size_t total = kGuard * 2 + b->clip_bytes * arena_clips;
  if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess) return nullptr;
  if (hipMalloc((void **)&b->arena, total) != hipSuccess) return nullptr;
  if (hipMemsetAsync(b->arena, 0, total, b->stream) != hipSuccess) return nullptr; // padding must read 0 forever (MD.cs:107)
  if (hipMalloc((void **)&b->d_fault, sizeof(int) * n_clips) != hipSuccess) return nullptr;
  if (hipMemsetAsync(b->d_fault, 0, sizeof(int) * n_clips, b->stream) != hipSuccess) return nullptr;
  if (hipEventCreate(&b->ev_begin) != hipSuccess || hipEventCreate(&b->ev_end) != hipSuccess) return nullptr;
  {
    // 64 rows: the 6-bit quantizer field of any descriptor stays inside; the intra kernel's tap table rides behind them
    std::vector<int32_t> tab((size_t)MOBI_SCALE_ROWS * MOBI_SCALE_STRIDE + MOBI_TAP_ENTRIES * 2 + 8, 0) /* + 8: the kernel fetches four entries at a time */;
    for (int q = 0; q < MOBI_SCALE_QMAX; q++) mobi_build_scale_table(q, &tab[(size_t)q * MOBI_SCALE_STRIDE]);
    mobi_build_scale_table(MOBI_SCALE_LITERAL, &tab[(size_t)MOBI_SCALE_LITERAL * MOBI_SCALE_STRIDE]); // the row of ones: literal frames (mobi_parse.cpp)
    if (!mobi_build_intra_taps((int16_t *)&tab[(size_t)MOBI_SCALE_ROWS * MOBI_SCALE_STRIDE], MOBI_TAP_PITCH)) {
      snprintf(g_last_hip_error, sizeof(g_last_hip_error), "intra tap table self-check failed");
      return nullptr;
    }
    if (hipMalloc((void **)&b->d_scale, tab.size() * 4) != hipSuccess) return nullptr;
    if (hipMemcpy(b->d_scale, tab.data(), tab.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  }
  {
    const size_t dbytes = (size_t)n_clips * b->g.mbw * b->g.mbh * 4;
    if (hipMalloc((void **)&b->d_done, dbytes) != hipSuccess) return nullptr;
    if (hipMemset(b->d_done, 0, dbytes) != hipSuccess) return nullptr;
    if (const char *fs = getenv("MOBI_FUSED_STEP_MBS")) b->fused_mbs = (size_t)strtoull(fs, nullptr, 10);
    // parse threads: measured on a 2 x 64-core host (tools/exp_hostparse.py, 1024 clips: 16 / 32 / 64 / 128 threads parse a P-frame step in
    // 20.0 / 10.3 / 5.5 / 4.3 ms): one per two hardware threads, 64 at most
    const int hw = (int)std::thread::hardware_concurrency();
    int helpers = std::min({n_clips, std::max(hw / 2, std::min(hw, 8)), 64}) - 1;
#if defined(MOBI_PROFILING)
    if (const char *hc = getenv("MOBI_HOST_CHUNK")) b->host_chunk = atoi(hc); // (tools/exp_hostparse.py; 0: no pipeline)
#endif
    if (const char *pt = getenv("MOBI_PARSE_THREADS")) helpers = atoi(pt) - 1;
    b->pool.reset(new ParsePool(std::max(0, std::min(helpers, 255))));
    // Where the parse runs by default: on the host threads up to ~20 resident clips per thread, on the GPU beyond (640x480, r04: 64 threads
    // take a step of 1024 / 1536 / 2048 clips in 9.1 / 13.6 / 18.2 ms, parse to planes; mobi_parse_frames in 13.2 / 10.6 / 10.4 ms)
    b->parse_mode = n_clips >= std::max(640, 20 * (b->pool->size() + 1));
    b->parse_auto = true;
    b->ls_policy = 2; // ... and which device parser is in front is decided step by step (ls_decide)
    if (const char *dp = getenv("MOBI_DEVICE_PARSE")) {
      const int v = atoi(dp); // 3: on the GPU, the lock-step parser in front of every step
      b->parse_mode = v == 3 ? 1 : std::max(0, std::min(2, v));
      b->ls_policy = v == 3 ? 1 : 0;
      b->parse_auto = false;
    }
  }
  if (b->debug == 9) {
    const size_t pbytes = (size_t)n_clips * (b->g.mbw * b->g.mbh) * 16;
    if (hipMalloc((void **)&b->d_prof, pbytes) != hipSuccess) return nullptr;
    if (hipMemset(b->d_prof, 0, pbytes) != hipSuccess) return nullptr;
  }
  if (hipStreamSynchronize(b->stream) != hipSuccess) return nullptr;
  return b.release();
}

void mobi_batch_destroy(mobi_batch *b) {
  if (!b) return;
  (void)hipSetDevice(b->device);
  delete b;
}

// DecodeFrame() of every clip with the bitstream parse on the GPU: upload Data[Offset..) of every clip, one parse launch
// (one wave per clip) that leaves descriptors, payload and intra lists in HBM, then the usual reconstruction launches.
//
// r05: WHICH side parses never shows in the result.  The device parsers finish the frames that decode without incident.  A frame they
// cannot finish -- every condition under which the reference throws, every walk through Internal[] (MD.cs:3424-3429), every value the
// command list has to escape (mobi_parse.cpp) -- is parsed again by the host parser INSIDE THE SAME CALL, from the decoder state the clip
// had when the frame started (the device keeps a ring of three states per clip: mobi_dparse.h, mobi_state.h), and its command list is
// written over the blank one the parse kernel left.  rc, Offset, Quantizer and the planes are then the host parser's, i.e. the same as
// at any other batch size; the clip stays with the host parser from there on (`on_host`), parsed beside the GPU's clips like the
// host-parsed share of the hybrid mode, which is the same mechanism with the last clips of the batch marked from the start.
//
// first use of the device-side parser: zeroed decoder state (a new MobiclipDecoder), result array, tables -- all of it or none of it
static int dp_init(mobi_batch *b) {
  const int n = b->n;
  if (b->d_pstate[0]) return MOBI_OK;
  auto init = [&]() -> int {
    HIP_TRY(hipMalloc((void **)&b->d_pres, sizeof(MobiDevResult) * n));
    HIP_TRY(hipMemset(b->d_pres, 0, sizeof(MobiDevResult) * n));
    std::vector<uint8_t> blob(MOBI_DT_BYTES);
    mobi_dparse_build_tables(b->version, blob.data());
    HIP_TRY(hipMalloc((void **)&b->d_ptables, MOBI_DT_BYTES));
    HIP_TRY(hipMemcpy(b->d_ptables, blob.data(), MOBI_DT_BYTES, hipMemcpyHostToDevice));
    if (int e = b->h_pres.reserve(sizeof(MobiDevResult) * n)) return e;
    HIP_TRY(hipMalloc((void **)&b->d_pstate_ls, sizeof(MobiDevState) * n));
    for (int k = 0; k < 3; k++) {
      HIP_TRY(hipMalloc((void **)&b->d_ptail[k], sizeof(MobiDevTail) * n));
      HIP_TRY(hipMemset(b->d_ptail[k], 0, sizeof(MobiDevTail) * n));
    }
    for (int k = 2; k >= 0; k--) { // entry 0 last: its presence means "initialised"
      HIP_TRY(hipMalloc((void **)&b->d_pstate[k], sizeof(MobiDevState) * n));
      HIP_TRY(hipMemset(b->d_pstate[k], 0, sizeof(MobiDevState) * n));
    }
    return MOBI_OK;
  };
  if (int e = init()) {
    if (b->d_pres) { (void)hipFree(b->d_pres); b->d_pres = nullptr; }
    if (b->d_ptables) { (void)hipFree(b->d_ptables); b->d_ptables = nullptr; }
    if (b->d_pstate_ls) { (void)hipFree(b->d_pstate_ls); b->d_pstate_ls = nullptr; }
    for (int k = 0; k < 3; k++) {
      if (b->d_pstate[k]) { (void)hipFree(b->d_pstate[k]); b->d_pstate[k] = nullptr; }
      if (b->d_ptail[k]) { (void)hipFree(b->d_ptail[k]); b->d_ptail[k] = nullptr; }
    }
    return e;
  }
  b->dev_quant.assign(n, 0);
  b->dev_yuvfmt.assign(n, 0);
  if ((int)b->on_host.size() != n) b->on_host.assign(n, 0);
  b->host_share.assign(n, 0);
  b->clean_run.assign(n, 0);
  b->clean_need.assign(n, 4);
  return MOBI_OK;
}
// stage [bit_off u64 x n][bit_len u32 x n][bits: each clip 8-byte aligned, zero padded] into pinned memory; the host parser's clips
// (b->on_host) carry MOBI_DP_SKIP for a length and no bits
struct DpStaged { size_t hdr_bytes = 0, bytes = 0, max_len = 0; int n_dev = 0, n_iframes = 0; }; // n_dev: clips the GPU parses; n_iframes: of them, I-frames (first bit)
// Which device parser is in front of this step (parse mode by default, r05; tools/exp_lsab.sh, tools/exp_dparse.py, 640x480, ms per step,
// mobi_parse_frames against the lock-step parser at its best number of clips per wave):
//   P-frames: 2048 clips 8.8 / 9.2, 4096: 11.2 / 13.7, 8192: 21.2 / 18.5, 16384: 42 / 23.5, 24576: 63 / 26.2
//   I-frames: 1024 clips 23.4 / 11.7, 2048: 30.0 / 11.3, 4096: 30.0 / 16.5, 8192: 54.9 / 23.6
// -- one wave per clip wins while every clip has a wave slot of its own and the frames are P-frames; an I-frame's macroblocks are all of one
// kind, which is what lanes in lock step like.
static bool ls_decide(const mobi_batch *b, const DpStaged &st) {
  if (b->ls_policy != 2) return b->ls_policy == 1;
  const bool iframe_step = 2 * st.n_iframes > st.n_dev;
  return st.n_dev >= 5120 || (iframe_step && st.n_dev >= 768);
}
// dev != nullptr: the image is also sent to *dev on `up`, chunk by chunk while the next chunk is gathered (one thread of the pool sits in
// the copy calls, which return when the bus is done: 8 ms for the 370 MB of a step of 24576 clips; the others gather) -- r04: gathering and
// sending were 12 of the 45 ms of such a step, one after the other.
static int dp_stage(mobi_batch *b, const uint8_t *const *data, const size_t *len, const int32_t *offsets, PinnedBuf &stage, DpStaged &st,
                    DevBuf *dev = nullptr, bool dev_headroom = false, hipStream_t up = nullptr) {
  const int n = b->n, nd = n, n_mbs = b->g.mbw * b->g.mbh;
  const auto t_stage0 = std::chrono::steady_clock::now();
  constexpr size_t kBitPad = 32; // the reader runs two 8-byte registers ahead
  std::vector<uint64_t> boff(nd);
  std::vector<uint32_t> blen(nd);
  size_t pos = 0, max_len = 0;
  const size_t frame_bound = (size_t)n_mbs * 4096 + 64;
  for (int i = 0; i < n; i++) {
    const int64_t o = offsets[i];
    size_t l = (data[i] && o >= 0 && (uint64_t)o < len[i]) ? len[i] - (size_t)o : 0; // nothing readable: the first ReadU16LE throws
    // MOC5 callers pass the whole file as Data (Form1.cs:292-302).  The reader advances two bytes per refill and refills at most
    // once per syntax element: <= ~1450 refills per macroblock (127 partition nodes, 384 levels of up to three reads each), so
    // bytes beyond 4 KB per macroblock cannot influence the parse of this frame
    l = std::min(l, frame_bound);
    max_len = std::max(max_len, l); // (the host parser's clips too: their share of the payload bound counts)
    boff[i] = pos;
    if (b->on_host[i]) { blen[i] = MOBI_DP_SKIP; continue; }
    blen[i] = (uint32_t)l;
    pos += align_up(l + kBitPad, 8);
    st.n_dev++;
    st.n_iframes += l >= 2 && (data[i][o + 1] & 0x80) != 0; // the frame's first bit: the top bit of its first 16-bit little-endian word (MD.cs:110-113)
  }
  pos += 64; // (a skipped clip's offset points at readable bytes too)
  const size_t hdr_bytes = align_up((size_t)nd * 12, 16);
  if (hdr_bytes + pos > stage.cap) // pinned memory is slow to allocate: leave room for the longer frames to come
    if (int e = stage.reserve(hdr_bytes + pos + (hdr_bytes + pos) / 4)) return e;
  uint8_t *hs = stage.p;
  memcpy(hs, boff.data(), (size_t)nd * 8);
  memcpy(hs + (size_t)nd * 8, blen.data(), (size_t)nd * 4);
  memset(hs + hdr_bytes + pos - 64, 0, 64);
  auto gather = [&](int i) {
    if (blen[i] == MOBI_DP_SKIP) return;
    uint8_t *dst = hs + hdr_bytes + boff[i];
    if (blen[i]) memcpy(dst, data[i] + offsets[i], blen[i]);
    memset(dst + blen[i], 0, align_up(blen[i] + kBitPad, 8) - blen[i]);
  };
  constexpr int kRun = 32; // frames per pool task (one per frame has the pool's threads queue at its counter: gop_begin measured it)
  if (!dev) {
    b->pool->run((nd + kRun - 1) / kRun, [&](int j) { for (int i = j * kRun, e = std::min(nd, i + kRun); i < e; i++) gather(i); });
  } else {
    const size_t need = hdr_bytes + pos;
    if (need > dev->cap) // growing frees and allocates (a device-wide stall): asynchronous steps leave room for the longer frames to come
      if (int e = dev->reserve(dev_headroom ? need + need / 4 : need)) return e;
    const int chunks = nd >= 2048 ? 4 : 1;
    std::atomic<int> up_err{0};
    auto end_of = [&](int c1) { return c1 < nd ? hdr_bytes + (size_t)boff[c1] : need; }; // byte offset where clip c1's bits start
    for (int k = 0; k <= chunks; k++) {
      const int c0 = k < chunks ? (int)((long)nd * k / chunks) : nd, c1 = k < chunks ? (int)((long)nd * (k + 1) / chunks) : nd;
      const int u0 = k >= 1 ? (int)((long)nd * (k - 1) / chunks) : 0, u1 = k >= 1 ? (int)((long)nd * k / chunks) : 0; // gathered in the round before
      const int n_up = u1 > u0 ? 1 : 0;
      if (n_up + (c1 - c0) == 0) continue;
      b->pool->run(n_up + (c1 - c0 + kRun - 1) / kRun, [&](int j) {
        if (j < n_up) {
          const size_t a = u0 == 0 ? 0 : end_of(u0), e = end_of(u1); // (the first chunk takes the header along)
          if (e > a && (hipSetDevice(b->device) != hipSuccess || hipMemcpyAsync(dev->p + a, hs + a, e - a, hipMemcpyHostToDevice, up) != hipSuccess)) up_err = 1;
          return;
        }
        for (int i = c0 + (j - n_up) * kRun, e = std::min(c1, i + kRun); i < e; i++) gather(i);
      });
    }
    if (up_err) return MOBI_E_DEVICE;
  }
  b->last_stage_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_stage0).count();
  st.hdr_bytes = hdr_bytes;
  st.bytes = hdr_bytes + pos;
  st.max_len = max_len;
  return MOBI_OK;
}
// output buffers + the parse launch.  A clip's payload can never exceed 448 words per macroblock, nor 64 per macroblock + one level
// per THREE bits read (a level word stands for a token with a level, and the shortest of those -- either table, either version -- has two
// bits and a sign: mobi_dparse_build_tables; tests/test_parse_fallback.py checks the tables for it).  `busy`: earlier steps may still be using the buffers (asynchronous steps): drain the stream before growing one.
// *state_in: which entry of the state ring the step reads (the state its frames start from).
struct DpOut { DevBuf *desc, *pay, *items; MobiDevResult *res; hipStream_t stream; bool async = false; };
static int dp_parse(mobi_batch *b, const uint8_t *d_bits, const DpStaged &st, bool busy, const DpOut &o, int *state_in) {
  const int n = b->n, n_mbs = b->g.mbw * b->g.mbh;
  // (the longest frame of a step varies from step to step: the bound follows it upwards in steps of a quarter, so that the payload
  // arena -- gigabytes for thousands of clips -- is not freed and allocated again every few frames)
  if (st.max_len > b->dp_len_hint) b->dp_len_hint = st.max_len + st.max_len / 4;
  const size_t cap_words = std::min<size_t>((size_t)n_mbs * 448 + MOBI_WIDE_PARAMS, (size_t)n_mbs * 64 + (8 * b->dp_len_hint + 2) / 3) + 448 + 64;
  // MbDesc.payload_off is relative to the clip's own part of the arena, which may then be as large as HBM lets it (24576 clips of
  // 640x480 need 13 G words for an I-frame); the host parser's clips write into their own parts like everybody else (dp_override)
  b->pay_clip_words = (uint32_t)cap_words;
  const size_t want_desc = align_up((size_t)n * n_mbs * sizeof(MbDesc) + 8 * sizeof(MbDesc), kAlign), want_pay = align_up((size_t)n * cap_words * 4 + kPaySlack, kAlign),
               want_items = (size_t)n * n_mbs * 4;
  if (busy && (want_desc > (*o.desc).cap || want_pay > (*o.pay).cap || want_items > (*o.items).cap)) HIP_TRY(hipStreamSynchronize(o.stream));
  if (int e = (*o.desc).reserve(want_desc)) return e;
  if (int e = (*o.pay).reserve(want_pay)) return e;
  if (int e = (*o.items).reserve(want_items)) return e;
  MobiDevParseArgs pa;
  memset(&pa, 0, sizeof(pa));
  pa.bits = d_bits + st.hdr_bytes;
  pa.bit_off = (const uint64_t *)d_bits;
  pa.bit_len = (const uint32_t *)(d_bits + (size_t)n * 8);
  pa.tables = b->d_ptables;
  const int in = b->ps_cur, out = (b->ps_cur + 1) % 3;
  pa.state_in = b->d_pstate[in]; pa.state_out = b->d_pstate[out];
  pa.tail_in = b->d_ptail[in]; pa.tail_out = b->d_ptail[out];
  pa.scale = b->d_scale;
  pa.state_ls = b->d_pstate_ls;
  pa.lockstep = b->lockstep ? (o.async ? 2 : 1) : 0; // (2: an asynchronous step; the launch has had one shape for both since r05: mobi_launch_parse_ls)
  pa.pay_local = 1;
  pa.desc = (MbDesc *)(*o.desc).p;
  pa.payload = (uint32_t *)(*o.pay).p;
  pa.items = (uint32_t *)(*o.items).p;
  pa.res = o.res;
  pa.pay_cap = (uint32_t)cap_words;
  b->last_pay_cap = cap_words;
  pa.n_clips = n; pa.version = b->version;
  pa.width = b->g.width; pa.height = b->g.height; pa.stride = b->g.stride; pa.lg = b->g.lg; pa.mbw = b->g.mbw; pa.mbh = b->g.mbh;
  if (b->ktiming && !b->ev_p0) { (void)hipEventCreate(&b->ev_p0); (void)hipEventCreate(&b->ev_p1); }
  const bool ptime = b->ktiming && b->ev_p0 && b->ev_p1;
  if (ptime) (void)hipEventRecord(b->ev_p0, o.stream);
  if (mobi_launch_parse(&pa, o.stream) != 0) return MOBI_E_DEVICE;
  if (ptime) (void)hipEventRecord(b->ev_p1, o.stream);
  b->ps_cur = out;
  *state_in = in;
  return MOBI_OK;
}

// The command lists of host-parsed clips of a device-parsed step (`clips`, ascending; b->cur[c] holds clip c's frame when rc[c] is MOBI_OK):
// staged in pinned memory and copied into the clips' own rows of what the parse kernels leave -- descriptor rows, payload parts, item
// rows, result records -- on stream s.  Runs of neighbouring clips (the hybrid mode's share) go as one copy per table.
struct DpRows { uint8_t *desc, *pay, *items; MobiDevResult *res; size_t cap_words; };
// frames != nullptr: frames[j] is clips[j]'s parsed frame (nullptr: it failed) and rcs[j] its rc -- both by POSITION in `clips` (the
// frame-parallel path: a clip has K frames); else b->cur[c] and rc[c] by clip.
static int dp_override(mobi_batch *b, const std::vector<int> &clips, const int *rc, PinnedBuf &stage, const DpRows &d, hipStream_t s,
                       const ParsedFrame *const *frames = nullptr, const int *rcs = nullptr) {
  const int k = (int)clips.size(), n_mbs = b->g.mbw * b->g.mbh;
  if (k == 0) return MOBI_OK;
  auto frame_of = [&](int j) -> const ParsedFrame * { return frames ? frames[j] : (rc[clips[j]] == MOBI_OK ? &b->cur[clips[j]] : nullptr); };
  auto rc_of = [&](int j) { return frames ? rcs[j] : rc[clips[j]]; };
  size_t pitch_w = 0; // payload words per clip in the staging area
  for (int j = 0; j < k; j++)
    if (const ParsedFrame *f = frame_of(j)) pitch_w = std::max(pitch_w, f->payload.size());
  pitch_w = align_up(pitch_w + 4, 4);
  if (pitch_w > d.cap_words) return MOBI_E_DEVICE; // cannot happen: cap_words bounds any clip's payload
  const size_t desc_b = (size_t)n_mbs * sizeof(MbDesc), item_b = (size_t)n_mbs * 4, res_b = sizeof(MobiDevResult), pay_b = pitch_w * 4;
  const size_t o_item = align_up((size_t)k * desc_b, 16), o_res = o_item + align_up((size_t)k * item_b, 16), o_pay = o_res + align_up((size_t)k * res_b, 16);
  if (int e = stage.reserve(o_pay + (size_t)k * pay_b)) return e;
  uint8_t *h = stage.p;
  const int groups = std::min(k, 32);
  b->pool->run(groups, [&](int g) {
    for (int j = (int)((long)k * g / groups), e = (int)((long)k * (g + 1) / groups); j < e; j++) {
      const ParsedFrame *f = frame_of(j);
      MbDesc *dd = (MbDesc *)(h + (size_t)j * desc_b);
      uint32_t *it = (uint32_t *)(h + o_item + (size_t)j * item_b);
      MobiDevResult *rr = (MobiDevResult *)(h + o_res + (size_t)j * res_b);
      memset(rr, 0, sizeof(*rr));
      rr->rc = rc_of(j);
      if (!f) { // a failed clip: descriptors typed "intra" that no launch list references
        for (int m = 0; m < n_mbs; m++) dd[m] = MbDesc{0, MOBI_MB_INTRA, 0, 0, 0, 0, 0, 0};
        continue;
      }
      memcpy(dd, f->desc.data(), desc_b); // (payload_off counts from the clip's own part of the arena, as the parser wrote it)
      if (!f->payload.empty()) memcpy(h + o_pay + (size_t)j * pay_b, f->payload.data(), f->payload.size() * 4);
      if (!f->intra_mbs.empty()) memcpy(it, f->intra_mbs.data(), f->intra_mbs.size() * 4); // level order: a valid order for the waits
      rr->n_intra = f->hdr.n_intra;
      rr->payload_words = f->hdr.payload_words;
      rr->quant = f->hdr.quantizer;
      rr->frame_type = f->hdr.frame_type;
    }
  });
  for (int j0 = 0; j0 < k;) {
    int j1 = j0 + 1;
    while (j1 < k && clips[j1] == clips[j1 - 1] + 1) j1++;
    const size_t c0 = (size_t)clips[j0], run = (size_t)(j1 - j0);
    HIP_TRY(hipMemcpyAsync(d.desc + c0 * desc_b, h + (size_t)j0 * desc_b, run * desc_b, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d.items + c0 * item_b, h + o_item + (size_t)j0 * item_b, run * item_b, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync((uint8_t *)(d.res + c0), h + o_res + (size_t)j0 * res_b, run * res_b, hipMemcpyHostToDevice, s));
    if (run == 1) HIP_TRY(hipMemcpyAsync(d.pay + c0 * d.cap_words * 4, h + o_pay + (size_t)j0 * pay_b, pay_b, hipMemcpyHostToDevice, s));
    else HIP_TRY(hipMemcpy2DAsync(d.pay + c0 * d.cap_words * 4, d.cap_words * 4, h + o_pay + (size_t)j0 * pay_b, pay_b, pay_b, run, hipMemcpyHostToDevice, s));
    j0 = j1;
  }
  return MOBI_OK;
}
// the decoder state the clips of `cl` had when the step that read ring entry `in` started, into their host parsers: two asynchronous copies per
// clip into pinned memory, ONE wait, the imports on the pool (r05 issued two blocking copies per clip: a step in which every clip of a large
// batch is handed over at once -- a ModsDS batch that cuts to a quantiser below 12 -- sat in tens of thousands of them)
static int dp_seed_parsers(mobi_batch *b, const std::vector<int> &cl, int in, PinnedBuf &stage, hipStream_t s) {
  if (cl.empty()) return MOBI_OK;
  const size_t rec = sizeof(MobiDevState) + sizeof(MobiDevTail);
  if (int e = stage.reserve(cl.size() * rec)) return e;
  for (size_t j = 0; j < cl.size(); j++) {
    HIP_TRY(hipMemcpyAsync(stage.p + j * rec, b->d_pstate[in] + cl[j], sizeof(MobiDevState), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(stage.p + j * rec + sizeof(MobiDevState), b->d_ptail[in] + cl[j], sizeof(MobiDevTail), hipMemcpyDeviceToHost, s));
  }
  HIP_TRY(hipStreamSynchronize(s));
  b->pool->run((int)cl.size(), [&](int j) {
    b->parsers[cl[j]]->import_state(*(const MobiDevState *)(stage.p + (size_t)j * rec), *(const MobiDevTail *)(stage.p + (size_t)j * rec + sizeof(MobiDevState)));
  });
  return MOBI_OK;
}

// Hybrid mode: how many clips (the last ones of the batch) the host pool parses beside the GPU.  (r05, tools/exp_hybrid.py: no share makes
// a step of up to 4096 clips faster than parse mode 1 -- 14.1 ms against 14.6 .. 16.2 -- because every clip has a wave slot of its own there
// and the parse launch is as long as ONE clip's parse whatever their number; the mode only pays beyond what the chip holds at once.)
static void hybrid_share(mobi_batch *b) {
  b->hybrid_host = std::min(1024, b->n / 5);
  if (const char *hh = getenv("MOBI_HYBRID_HOST_CLIPS")) b->hybrid_host = std::max(0, std::min(b->n - 1, atoi(hh)));
}

// A clip the host parser took over goes BACK to the device parsers once its frames are again the kind they finish: `clean_need`
// consecutive frames without a walk, a token without a level or a value beyond the device parsers' fields (MobiStreamParser::device_ready),
// doubled with every hand-over (4, 8, ... 256: a clip whose every frame needs the host parser -- a ModsDS stream below quantiser 12 -- costs
// a wasted device parse every few hundred frames, a glitch in an otherwise clean stream a handful of host-parsed frames).  The parser's state
// goes into entry `entry` of the state ring -- the one the next step's parse reads -- on stream s, through `stage`.
static int dp_return_list(mobi_batch *b, const std::vector<int> &back, int entry, PinnedBuf &stage, hipStream_t s);
static int dp_return(mobi_batch *b, const std::vector<int> &host_clips, const int *rc, int entry, PinnedBuf &stage, hipStream_t s) {
  std::vector<int> back;
  for (int c : host_clips) {
    if (b->host_share[c]) continue;
    if (rc[c] == MOBI_OK && b->parsers[c]->device_ready()) b->clean_run[c]++;
    else b->clean_run[c] = 0;
    if (b->clean_run[c] >= b->clean_need[c]) back.push_back(c);
  }
  return dp_return_list(b, back, entry, stage, s);
}
static int dp_return_list(mobi_batch *b, const std::vector<int> &back, int entry, PinnedBuf &stage, hipStream_t s) {
  if (back.empty()) return MOBI_OK;
  const size_t rec = sizeof(MobiDevState) + sizeof(MobiDevTail);
  if (int e = stage.reserve(back.size() * rec)) return e;
  for (size_t k = 0; k < back.size(); k++) {
    const int c = back[k];
    MobiDevState *st = (MobiDevState *)(stage.p + k * rec);
    MobiDevTail *tail = (MobiDevTail *)(stage.p + k * rec + sizeof(MobiDevState));
    b->parsers[c]->export_state(*st, *tail);
    HIP_TRY(hipMemcpyAsync(b->d_pstate[entry] + c, st, sizeof(*st), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(b->d_ptail[entry] + c, tail, sizeof(*tail), hipMemcpyHostToDevice, s));
    b->on_host[c] = 0;
    b->clean_run[c] = 0;
    b->returns++;
  }
  return MOBI_OK;
}

// Once the parsers have consumed a frame and the ring has turned, a call that fails before its reconstruction is complete must not leave
// any clip reporting MOBI_OK for a frame that was never reconstructed (host-parsed and device-parsed steps alike: ADVICE r03)
struct FailAll {
  int *rc; int n; hipStream_t stream = nullptr; bool armed = true;
  ~FailAll() {
    if (!armed) return;
    for (int i = 0; i < n; i++) if (rc[i] == MOBI_OK) rc[i] = MOBI_E_DEVICE;
    //  copies out of the pinned staging buffers may still be in flight on the error paths: the next call refills those buffers
    if (stream) (void)hipStreamSynchronize(stream);
  }
};

static int decode_device_parse(mobi_batch *b, const uint8_t *const *data, const size_t *len, int32_t *offsets, int *rc) {
  const int n = b->n;
  if (b->version != MOBI_VERSION_MODSDS && b->version != MOBI_VERSION_MOFLEX3DS) {
    for (int i = 0; i < n; i++) rc[i] = MOBI_E_VERSION;
    return MOBI_OK; // DecodeFrame() returns before touching the ring (MD.cs:56-61)
  }
  if (b->g.mbw > 64 || b->async_count || b->gop_count) return MOBI_E_ARG; // (asynchronous steps / groups in flight: wait for them first)
  if (int e = dp_init(b)) return e;
  if (b->hybrid_host && b->frames_started == 0) // hybrid: the last clips are the host parsers' from the start (their state is a new decoder's: nothing to seed)
    for (int i = n - b->hybrid_host; i < n; i++) b->on_host[i] = b->host_share[i] = 1;
  std::vector<int> host_clips;
  for (int i = 0; i < n; i++)
    if (b->on_host[i]) host_clips.push_back(i);
  DpStaged st;
  using clk = std::chrono::steady_clock;
  auto ms_since = [](clk::time_point x) { return std::chrono::duration<float, std::milli>(clk::now() - x).count(); };
  const auto q0 = clk::now();
  if (int e = dp_stage(b, data, len, offsets, b->h_stage, st, &b->d_bits, false, b->stream)) return e; // gathered and on their way
  b->phase_ms[0] = ms_since(q0);
  b->lockstep = ls_decide(b, st);
  int state_in = 0;
  if (int e = dp_parse(b, b->d_bits.p, st, false, DpOut{&b->d_pdesc, &b->d_ppay, &b->d_pitems, b->d_pres, b->stream}, &state_in)) return e;
  b->phase_ms[1] = ms_since(q0);
  const size_t cap_words = b->last_pay_cap;
  const DpRows rows{b->d_pdesc.p, b->d_ppay.p, b->d_pitems.p, b->d_pres, cap_words};
  const bool ptime = b->ktiming && b->ev_p0 && b->ev_p1;
  MobiDevResult *res = (MobiDevResult *)b->h_pres.p;
  HIP_TRY(hipMemcpyAsync(res, b->d_pres, sizeof(MobiDevResult) * n, hipMemcpyDeviceToHost, b->stream));
  static const uint8_t kNoData2[2] = {0, 0};
  auto host_parse = [&](const std::vector<int> &cl) {
    b->pool->run((int)cl.size(), [&](int j) { const int i = cl[j]; rc[i] = b->parsers[i]->parse_frame(data[i] ? data[i] : kNoData2, data[i] ? len[i] : 0, &offsets[i], b->cur[i]); });
  };
  if (!host_clips.empty()) { // while the GPU parses its clips, the host pool parses its own and sends their command lists up beside it
    if (!b->stream2) HIP_TRY(hipStreamCreateWithFlags(&b->stream2, hipStreamNonBlocking));
    if (!b->ev_up) HIP_TRY(hipEventCreateWithFlags(&b->ev_up, hipEventDisableTiming));
    host_parse(host_clips);
    if (int e = dp_override(b, host_clips, rc, b->h_stage2, rows, b->stream2)) return e; // (no kernel touches these clips' rows: MOBI_DP_SKIP)
    HIP_TRY(hipEventRecord(b->ev_up, b->stream2));
    HIP_TRY(hipStreamWaitEvent(b->stream, b->ev_up, 0));
    HIP_TRY(hipStreamSynchronize(b->stream2));
  }
  HIP_TRY(hipStreamSynchronize(b->stream)); // the launch sizes below depend on what the parse found
  b->phase_ms[2] = ms_since(q0);
  if (ptime) { float ms = 0; if (hipEventElapsedTime(&ms, b->ev_p0, b->ev_p1) == hipSuccess) b->last_parse_ms = ms; }
  // frames the device parsers could not finish: the host parser takes them over from the state they started with
  std::vector<int> fb;
  for (int i = 0; i < n; i++)
    if (!b->on_host[i] && res[i].rc != MOBI_OK) fb.push_back(i);
  if (!fb.empty()) {
    if (int e = dp_seed_parsers(b, fb, state_in, b->h_ret, b->stream)) return e; // (the stream is idle: it was waited for above)
    host_parse(fb);
    for (int c : fb) {
      b->on_host[c] = 1;
      b->clean_run[c] = 0;
      b->clean_need[c] = (uint16_t)std::min(256, 2 * (int)b->clean_need[c]);
      host_clips.push_back(c);
    }
    b->fallbacks += fb.size();
    if (int e = dp_override(b, fb, rc, b->h_stage2, rows, b->stream)) return e; // (behind the parse kernels, which blanked these clips' rows)
  }
  uint32_t K = 0;
  b->ls_finished = b->lockstep ? 0 : -1;
  for (int i = 0; i < n; i++) {
    if (b->on_host[i]) {
      b->dev_quant[i] = b->parsers[i]->quantizer();
      b->dev_yuvfmt[i] = b->parsers[i]->yuv_format();
      if (rc[i] == MOBI_OK) K = std::max(K, b->cur[i].hdr.n_intra);
      continue;
    }
    if (b->lockstep && res[i].pad == MOBI_LS_MAGIC) b->ls_finished++;
    rc[i] = res[i].rc;
    offsets[i] += res[i].consumed;
    b->dev_quant[i] = res[i].quant;
    b->dev_yuvfmt[i] = res[i].yuvfmt;
    K = std::max(K, res[i].n_intra);
  }
  // 3. reconstruction straight from what the parse left in HBM
  b->ring_base = (b->ring_base + 1) % 6; // Y[i] = Y[i-1]; Y[0] = new (MD.cs:102-108) -- even if the parse threw
  b->step_tag = b->step_tag + 1 ? b->step_tag + 1 : 1;
  b->argb_all_valid = false;
  b->frames_started++;
  FailAll fail_all{rc, n, b->stream}; // rc[], Offset, the ring and the decoder state have advanced: the launches below must complete
  MobiReconArgs a = b->args(b->d_pdesc.p, b->d_ppay.p);
  a.pay_clip_words = b->pay_clip_words;
  a.done = b->d_done;
  if (mobi_launch_inter(&a, b->stream) != 0) return MOBI_E_DEVICE;
  if (K && mobi_launch_intra_cl(&a, (const uint32_t *)b->d_pitems.p, &b->d_pres[0].n_intra, (int)(sizeof(MobiDevResult) / 4), (int)K, 0, b->stream) != 0)
    return MOBI_E_DEVICE;
  HIP_TRY(hipMemcpyAsync(b->h_fault.data(), b->d_fault, sizeof(int) * n, hipMemcpyDeviceToHost, b->stream));
  HIP_TRY(hipMemsetAsync(b->d_fault, 0, sizeof(int) * n, b->stream));
  if (!host_clips.empty())
    if (int e = dp_return(b, host_clips, rc, b->ps_cur, b->h_ret, b->stream)) return e; // (ps_cur: what the next step's parse reads)
  b->phase_ms[3] = ms_since(q0);
  HIP_TRY(hipStreamSynchronize(b->stream));
  b->phase_ms[4] = ms_since(q0);
  fail_all.armed = false;
  for (int i = 0; i < n; i++)
    if (rc[i] == MOBI_OK && b->h_fault[i]) rc[i] = (b->h_fault[i] & 2) ? MOBI_E_DEVICE : MOBI_E_CLAMP;
  return MOBI_OK;
}

// Frames of SOME clips through the reconstruction kernels on their own, into the ring slot `ring_base` names: the repair of clips whose
// asynchronous step had been enqueued before anybody knew that their frame was not the device parser's to finish (mobi_batch_wait).
// `clips` ascending, frames[j] = clips[j]'s parsed frame (nullptr: the host parser rejected it: nothing is written).  Neighbouring clips
// form a run, and a run is one compact frame step of its own -- its descriptor table, its payload arena, its launch items, planes and
// tags offset to its first clip -- so "every clip of the batch at once" is ONE set of launches and a lone glitch a set of its own.  One
// upload, one read-back of the fault words, one wait for the lot (r05: a parse, an upload, launches and a wait PER CLIP, on the calling thread).
static int recon_clips(mobi_batch *b, const std::vector<int> &clips, const ParsedFrame *const *frames, int ring_base, int *fault_out) {
  const int n_mbs = b->g.mbw * b->g.mbh, k = (int)clips.size();
  if (k == 0) return MOBI_OK;
  struct Run { int j0, j1; size_t desc_off, pay_off, item_off; LevelPlan plan; };
  std::vector<Run> runs;
  size_t total = 0;
  for (int j0 = 0; j0 < k;) {
    int j1 = j0 + 1;
    while (j1 < k && clips[j1] == clips[j1 - 1] + 1) j1++;
    Run r;
    r.j0 = j0; r.j1 = j1;
    std::vector<const ParsedFrame *> fr(frames + j0, frames + j1);
    r.plan.build(fr, b->g.mbw);
    const size_t desc_bytes = align_up((size_t)(j1 - j0) * n_mbs * sizeof(MbDesc) + 8 * sizeof(MbDesc), kAlign), pay_bytes = align_up(step_payload_words(fr) * 4 + kPaySlack, kAlign);
    if (step_payload_words(fr) + kPaySlack / 4 >= ((uint64_t)1 << 32)) return MOBI_E_ARG;
    r.desc_off = total; r.pay_off = total + desc_bytes; r.item_off = r.pay_off + pay_bytes;
    total = r.item_off + align_up(r.plan.items.size() * 4 + 16, kAlign);
    runs.push_back(std::move(r));
    j0 = j1;
  }
  if (int e = b->h_fix.reserve(total)) return e;
  if (int e = b->d_fix.reserve(total)) return e;
  b->pool->run((int)runs.size(), [&](int i) {
    const Run &r = runs[i];
    std::vector<const ParsedFrame *> fr(frames + r.j0, frames + r.j1);
    memset(b->h_fix.p + r.desc_off, 0, r.item_off - r.desc_off);
    step_write(fr, n_mbs, (MbDesc *)(b->h_fix.p + r.desc_off), (uint32_t *)(b->h_fix.p + r.pay_off));
    if (!r.plan.items.empty()) memcpy(b->h_fix.p + r.item_off, r.plan.items.data(), r.plan.items.size() * 4);
  });
  HIP_TRY(hipMemcpyAsync(b->d_fix.p, b->h_fix.p, total, hipMemcpyHostToDevice, b->stream));
  b->step_tag = b->step_tag + 1 ? b->step_tag + 1 : 1; // (one tag for all runs: they touch different clips' tags)
  for (const Run &r : runs) {
    const int c0 = clips[r.j0], m = r.j1 - r.j0;
    MobiReconArgs a = b->args(b->d_fix.p + r.desc_off, b->d_fix.p + r.pay_off);
    a.planes += (size_t)c0 * b->clip_bytes;
    a.n_clips = m;
    a.fault = b->d_fault + c0;
    a.done = b->d_done + (size_t)c0 * n_mbs;
    a.ring_base = ring_base;
    // (two launches, never the one-launch step: its give-up path -- fault bit 2, a dispatch order never seen -- has no retry here)
    if (int e = b->launch_plan(a, r.plan, (const uint32_t *)(b->d_fix.p + r.item_off), m, false)) return e;
  }
  HIP_TRY(hipMemcpyAsync(b->h_fault.data(), b->d_fault, sizeof(int) * b->n, hipMemcpyDeviceToHost, b->stream));
  for (const Run &r : runs) HIP_TRY(hipMemsetAsync(b->d_fault + clips[r.j0], 0, sizeof(int) * (r.j1 - r.j0), b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));
  b->drain_events();
  for (int j = 0; j < k; j++) fault_out[j] = b->h_fault[clips[j]];
  return MOBI_OK;
}

// ---- asynchronous frame steps (device parse) -------------------------------------------------------------------------------
// mobi_batch_decode is DecodeFrame(): it returns when the frame is there.  A caller that already holds the next frame of every clip
// (demuxed Moflex / Mods packets: Offset does not depend on the previous frame's parse) can keep two steps in flight instead:
// submit gathers the bytes into pinned memory, uploads them on a second stream and enqueues parse + reconstruction behind the
// previous step without a host round trip in between (the intra launch covers every slot a clip could use; workgroups past a
// clip's count leave at once); wait hands out rc[] / Offset of the oldest step when its reconstruction is done.  The host gathers
// step n + 1 while the GPU parses step n, and the GPU never waits for the host between parse and reconstruction.
// A frame the device parser cannot finish shows up in wait: the clip's frame (and the next one, if a second step is in flight) is then
// parsed by the host parser and reconstructed on its own there (async_repair), and the clip is the host parser's in later submits.
int mobi_batch_submit(mobi_batch *b, const uint8_t *const *data, const size_t *len, const int32_t *offsets) {
  if (!b || !data || !len || !offsets) return MOBI_E_ARG;
  HIP_TRY(hipSetDevice(b->device));
  const int n = b->n, n_mbs = b->g.mbw * b->g.mbh;
  if (b->frames_started == 0 && b->async_seq == 0) { // an asynchronous batch parses on the GPU from its first frame
    if (b->parse_auto) { b->parse_mode = 1; b->parse_auto = false; }
    if (b->parse_mode == 2) hybrid_share(b);
  }
  if (b->poisoned) return MOBI_E_DEVICE;
  if (b->parse_mode == 0) return MOBI_E_ARG; // the decoder state of this batch lives in the host parsers
  if (b->version != MOBI_VERSION_MODSDS && b->version != MOBI_VERSION_MOFLEX3DS) return MOBI_E_VERSION;
  if (b->g.mbw > 64 || b->async_count >= 2 || b->gop_count) return MOBI_E_ARG;
  if (int e = dp_init(b)) return e;
  if (b->hybrid_host && b->frames_started == 0)
    for (int i = n - b->hybrid_host; i < n; i++) b->on_host[i] = b->host_share[i] = 1;
  mobi_batch::AsyncSlot &S = b->aslot[(b->async_head + b->async_count) & 1];
  if (!S.ev_up) {
    HIP_TRY(hipEventCreateWithFlags(&S.ev_up, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&S.ev_done, hipEventDisableTiming));
  }
  if (!S.ev_parsed) HIP_TRY(hipEventCreateWithFlags(&S.ev_parsed, hipEventDisableTiming));
  if (!b->stream2) HIP_TRY(hipStreamCreateWithFlags(&b->stream2, hipStreamNonBlocking));
  if (!b->stream_p) HIP_TRY(hipStreamCreateWithFlags(&b->stream_p, hipStreamNonBlocking));
  DpStaged st;
  if (int e = dp_stage(b, data, len, offsets, S.h_stage, st)) return e; // (this slot's previous step was waited for: its upload is done)
  if (int e = S.d_pres.reserve(sizeof(MobiDevResult) * n)) return e;
  if (st.bytes > S.d_bits.cap) // growing frees and allocates (a device-wide stall): leave room for the longer frames to come
    if (int e = S.d_bits.reserve(st.bytes + st.bytes / 4)) return e;
  if (int e = S.h_pres.reserve(sizeof(MobiDevResult) * n)) return e;
  if (int e = S.h_fault.reserve(sizeof(int) * n)) return e;
  S.offs.assign(offsets, offsets + n);
  S.n_dev = n;
  b->lockstep = S.lockstep = ls_decide(b, st);
  S.hdr_bytes = st.hdr_bytes;
  S.is_host.assign(n, 0);
  S.host_rc.assign(n, MOBI_OK);
  S.host_off.assign(n, 0);
  S.host_quant.assign(n, 0);
  S.host_yuv.assign(n, 0);
  // the host parser's clips: parsed now (the caller's bytes are only good during this call), their command lists staged for the copy below
  std::vector<int> host_clips;
  for (int i = 0; i < n; i++)
    if (b->on_host[i]) host_clips.push_back(i);
  if (!host_clips.empty()) {
    static const uint8_t kNoData2[2] = {0, 0};
    for (int i : host_clips) { S.is_host[i] = 1; S.host_off[i] = offsets[i]; }
    b->pool->run((int)host_clips.size(), [&](int j) {
      const int i = host_clips[j];
      S.host_rc[i] = b->parsers[i]->parse_frame(data[i] ? data[i] : kNoData2, data[i] ? len[i] : 0, &S.host_off[i], b->cur[i]);
      S.host_quant[i] = b->parsers[i]->quantizer();
      S.host_yuv[i] = b->parsers[i]->yuv_format();
    });
  }
  // Everything that can fail without touching the device is behind us.  From the first enqueue on, a failure leaves work in flight
  // that reads this slot's pinned memory and (later) a ring that has turned without a step to wait for: the batch is drained and
  // POISONED -- every later call reports MOBI_E_DEVICE -- rather than left in a state where the next submit reuses the slot.
  struct Poison {
    mobi_batch *b; bool armed = true;
    ~Poison() {
      if (!armed) return;
      if (b->stream2) (void)hipStreamSynchronize(b->stream2);
      if (b->stream_p) (void)hipStreamSynchronize(b->stream_p);
      (void)hipStreamSynchronize(b->stream);
      b->poisoned = true;
    }
  } poison{b};
  HIP_TRY(hipMemcpyAsync(S.d_bits.p, S.h_stage.p, st.bytes, hipMemcpyHostToDevice, b->stream2)); // beside whatever the step before is doing
  HIP_TRY(hipEventRecord(S.ev_up, b->stream2));
  // Three streams: the upload (stream2), the parse (stream_p: after the upload; the decoder state goes from parse to parse in its order)
  // and the reconstruction (stream: after this step's parse; the ring goes from step to step in its order).  The parse of this step
  // runs under the reconstruction of the one before -- the lock-step parser leaves most of the chip idle -- because what it writes
  // (descriptors, payload, intra lists, results) belongs to the slot, whose previous step has been waited for.
  // (Only the lock-step parser gets the stream of its own: the one-wave-per-clip parser fills every wave slot of the chip with 122-register
  // waves, and a reconstruction launched beside it waits for them to leave anyway -- 20.5 instead of 13.2 ms per step of 4096 clips.)
  hipStream_t ps = b->lockstep ? b->stream_p : b->stream;
  HIP_TRY(hipStreamWaitEvent(ps, S.ev_up, 0));
  { // the parser in front is chosen step by step (ls_decide), and with it the stream: the state this parse reads is the one the step before wrote
    mobi_batch::AsyncSlot &prev = b->aslot[(b->async_head + b->async_count + 1) & 1];
    if (b->async_seq > 0 && prev.ev_parsed && prev.parsed_recorded) HIP_TRY(hipStreamWaitEvent(ps, prev.ev_parsed, 0));
  }
  MobiDevResult *d_res = (MobiDevResult *)S.d_pres.p;
  if (int e = dp_parse(b, S.d_bits.p, st, false, DpOut{&S.d_pdesc, &S.d_ppay, &S.d_pitems, d_res, ps, true}, &S.state_in)) return e;
  if (!host_clips.empty()) {
    if (int e = dp_override(b, host_clips, S.host_rc.data(), S.h_over, DpRows{S.d_pdesc.p, S.d_ppay.p, S.d_pitems.p, d_res, b->last_pay_cap}, ps)) return e;
    if (int e = dp_return(b, host_clips, S.host_rc.data(), b->ps_cur, S.h_ret, ps)) return e; // (the parse kernels skip these clips: their entries are free to write)
  }
  HIP_TRY(hipMemcpyAsync(S.h_pres.p, d_res, sizeof(MobiDevResult) * n, hipMemcpyDeviceToHost, ps));
  HIP_TRY(hipEventRecord(S.ev_parsed, ps));
  S.parsed_recorded = true;
  if (ps != b->stream) HIP_TRY(hipStreamWaitEvent(b->stream, S.ev_parsed, 0));
  // reconstruction straight from what the parse leaves in HBM (failed clips: blank descriptors, no items)
  b->ring_base = (b->ring_base + 1) % 6; // Y[i] = Y[i-1]; Y[0] = new (MD.cs:102-108) -- even if the parse throws
  b->step_tag = b->step_tag + 1 ? b->step_tag + 1 : 1;
  b->argb_all_valid = false;
  b->frames_started++;
  S.ring_base = b->ring_base;
  MobiReconArgs a = b->args(S.d_pdesc.p, S.d_ppay.p);
  a.pay_clip_words = b->pay_clip_words;
  a.done = b->d_done;
  if (mobi_launch_inter(&a, b->stream) != 0) return MOBI_E_DEVICE;
  // the parse has not run yet, so nobody knows how many intra macroblocks the longest list will have: MOBI_ASYNC_INTRA_SLOTS slots are launched
  // (a P-frame's lists are shorter), and a second launch (mobi_recon_intra_walk, one workgroup per four clips) goes through the rest of theirs (an I-frame: every macroblock)
  if (mobi_launch_intra_cl(&a, (const uint32_t *)S.d_pitems.p, &d_res[0].n_intra, (int)(sizeof(MobiDevResult) / 4), std::min(n_mbs, MOBI_ASYNC_INTRA_SLOTS), 1, b->stream) != 0) return MOBI_E_DEVICE;
  HIP_TRY(hipMemcpyAsync(S.h_fault.p, b->d_fault, sizeof(int) * n, hipMemcpyDeviceToHost, b->stream));
  HIP_TRY(hipMemsetAsync(b->d_fault, 0, sizeof(int) * n, b->stream));
  HIP_TRY(hipEventRecord(S.ev_done, b->stream));
  poison.armed = false;
  b->async_count++;
  b->async_seq++;
  return MOBI_OK;
}
// Clips of the oldest step in flight whose frame the device parser could not finish.  Everything enqueued is drained first (a second step
// in flight parsed these clips from a state the failed frame left behind: its frame is repaired as well); then, per clip: the state the
// frame started from goes into the host parser, the staged bitstream is parsed again, and the frame is reconstructed on its own into the
// ring slot its step wrote.  The results replace the device's in the slots' host-side tables.
static int async_repair(mobi_batch *b, mobi_batch::AsyncSlot &S, const std::vector<int> &failed) {
  if (b->stream2) HIP_TRY(hipStreamSynchronize(b->stream2));
  if (b->stream_p) HIP_TRY(hipStreamSynchronize(b->stream_p));
  HIP_TRY(hipStreamSynchronize(b->stream));
  mobi_batch::AsyncSlot *S1 = b->async_count == 2 ? &b->aslot[(b->async_head + 1) & 1] : nullptr;
  const int n = b->n, k = (int)failed.size(), nT = S1 ? 2 : 1;
  mobi_batch::AsyncSlot *const T[2] = {&S, S1};
  // r06: a batch operation.  The start states in one go, the parses on the pool (a clip's two frames in order), then per affected step ONE
  // reconstruction of the failed clips (recon_clips) -- r05 did all of it clip by clip on the calling thread: a GOP damaged in every clip of a
  // batch, or a ModsDS batch that cuts to a quantiser below 12 in one frame, stalled mobi_batch_wait for ~1.5 ms x 2 x clips.
  if (int e = dp_seed_parsers(b, failed, S.state_in, S.h_ret, b->stream)) return e;
  if (b->gop_frames.size() < (size_t)k * 2) b->gop_frames.resize((size_t)k * 2);
  std::vector<int> prc((size_t)k * 2, MOBI_OK);
  b->pool->run(k, [&](int j) {
    const int c = failed[j];
    for (int t = 0; t < nT; t++) {
      const uint64_t boff = ((const uint64_t *)T[t]->h_stage.p)[c];
      const uint32_t blen = ((const uint32_t *)(T[t]->h_stage.p + (size_t)n * 8))[c];
      int32_t off = 0;
      prc[(size_t)j * 2 + t] = b->parsers[c]->parse_frame(T[t]->h_stage.p + T[t]->hdr_bytes + boff, blen, &off, b->gop_frames[(size_t)j * 2 + t]);
      T[t]->is_host[c] = 1;
      T[t]->host_off[c] = T[t]->offs[c] + off;
      T[t]->host_quant[c] = b->parsers[c]->quantizer();
      T[t]->host_yuv[c] = b->parsers[c]->yuv_format();
    }
  });
  std::vector<const ParsedFrame *> fr(k);
  std::vector<int> fault(k);
  for (int t = 0; t < nT; t++) {
    for (int j = 0; j < k; j++) fr[j] = prc[(size_t)j * 2 + t] == MOBI_OK ? &b->gop_frames[(size_t)j * 2 + t] : nullptr;
    if (int e = recon_clips(b, failed, fr.data(), T[t]->ring_base, fault.data())) return e;
    for (int j = 0; j < k; j++) {
      int rc = prc[(size_t)j * 2 + t];
      if (rc == MOBI_OK && fault[j]) rc = (fault[j] & 2) ? MOBI_E_DEVICE : MOBI_E_CLAMP;
      T[t]->host_rc[failed[j]] = rc;
      ((int *)T[t]->h_fault.p)[failed[j]] = 0;
    }
  }
  for (int c : failed) {
    b->on_host[c] = 1;
    b->clean_run[c] = 0;
    b->clean_need[c] = (uint16_t)std::min(256, 2 * (int)b->clean_need[c]);
  }
  b->fallbacks += failed.size();
  return MOBI_OK;
}
int mobi_batch_wait(mobi_batch *b, int32_t *offsets_out, int *rc) {
  if (!b || !rc) return MOBI_E_ARG;
  HIP_TRY(hipSetDevice(b->device));
  if (b->poisoned) return MOBI_E_DEVICE;
  if (b->async_count == 0) return MOBI_E_ARG;
  mobi_batch::AsyncSlot &S = b->aslot[b->async_head & 1];
  HIP_TRY(hipEventSynchronize(S.ev_done));
  const MobiDevResult *res = (const MobiDevResult *)S.h_pres.p;
  const int *fault = (const int *)S.h_fault.p;
  std::vector<int> failed;
  for (int i = 0; i < S.n_dev; i++)
    if (!S.is_host[i] && res[i].rc != MOBI_OK) failed.push_back(i);
  if (!failed.empty())
    if (int e = async_repair(b, S, failed)) { b->poisoned = true; return e; }
  b->ls_finished = S.lockstep ? 0 : -1; // (of the step being reported, not of the one submitted last: ADVICE r05)
  for (int i = 0; i < S.n_dev; i++) {
    if (S.is_host[i]) {
      rc[i] = S.host_rc[i];
      if (offsets_out) offsets_out[i] = S.host_off[i];
      b->dev_quant[i] = S.host_quant[i];
      b->dev_yuvfmt[i] = S.host_yuv[i];
    } else {
      if (S.lockstep && res[i].pad == MOBI_LS_MAGIC) b->ls_finished++;
      rc[i] = res[i].rc;
      if (offsets_out) offsets_out[i] = S.offs[i] + (int32_t)res[i].consumed;
      b->dev_quant[i] = res[i].quant;
      b->dev_yuvfmt[i] = res[i].yuvfmt;
    }
    if (rc[i] == MOBI_OK && fault[i]) rc[i] = (fault[i] & 2) ? MOBI_E_DEVICE : MOBI_E_CLAMP;
  }
  b->async_head ^= 1;
  b->async_count--;
  return MOBI_OK;
}

// ---- frame-parallel groups (mobi_gop.h) -------------------------------------------------------------------------------------------
// K consecutive frames of every clip in one call: data / len / offsets / rc are [k * n_clips + c].  The frames are parsed SIDE BY SIDE --
// n * K virtual clips for the parse kernels, whose cost per frame falls with the number of lanes they are given -- and
// reconstructed as K steps in order.  begin: gather + upload (and, when nothing else is in flight, the parse); finish: the host parser's
// share and every hand-over, the K reconstruction steps, and the parse of the group begun behind it.
// Everything a group's parse and reconstruction write, sized when the group is BEGUN: a group that does not fit is refused by
// mobi_batch_gop_begin, with nothing enqueued (the parse of a group begun behind another goes out inside that one's mobi_batch_gop_finish:
// an allocation failing there would leave rings and parsers out of step).
static int gop_reserve(mobi_batch *b, mobi_batch::GopSlot &S) {
  const int n = b->n, K = S.K, n_mbs = b->g.mbw * b->g.mbh;
  const size_t nv = (size_t)n * K;
  if (S.max_len > b->dp_len_hint) b->dp_len_hint = S.max_len + S.max_len / 4;
  const size_t cap_words = std::min<size_t>((size_t)n_mbs * 448 + MOBI_WIDE_PARAMS, (size_t)n_mbs * 64 + (8 * b->dp_len_hint + 2) / 3) + 448 + 64; // (dp_parse has the bound's reasons)
  S.cap_words = cap_words;
  if (int e = S.d_desc.reserve(align_up(nv * n_mbs * sizeof(MbDesc) + 8 * sizeof(MbDesc), kAlign))) return e;
  if (int e = S.d_pay.reserve(align_up(nv * cap_words * 4 + kPaySlack, kAlign))) return e;
  if (int e = S.d_items.reserve(nv * n_mbs * 4)) return e;
  if (int e = S.d_res.reserve(nv * sizeof(MobiDevResult))) return e;
  if (int e = S.d_sin.reserve(nv * sizeof(MobiDevState))) return e;
  if (int e = S.d_sout.reserve(nv * sizeof(MobiDevState))) return e;
  if (int e = S.d_sls.reserve(nv * sizeof(MobiDevState))) return e;
  if (int e = S.d_tails.reserve(nv * sizeof(MobiDevTail))) return e;
  if (int e = S.d_fault.reserve(nv * sizeof(int))) return e;
  if (b->g.width < b->g.stride) { // the wavefront-ordered intra items (mobi_launch_gop_sort), for the most a group can hold: sized here, not when the
    // counts are known -- mobi_batch_gop_finish would sit in a hipFree + hipMalloc with the GPU idle whenever a group held more than the last
    if (int e = S.d_sorted.reserve((nv * n_mbs + (size_t)K * (3 * MOBI_SORT_LEVELS + 8)) * 16)) return e;
    if (int e = S.d_hist.reserve((size_t)3 * MOBI_GOP_PARSE_MAX * MOBI_SORT_LEVELS * 4)) return e;
  }
  if (int e = S.h_res.reserve(nv * sizeof(MobiDevResult))) return e;
  if (int e = S.h_fault.reserve(nv * sizeof(int))) return e;
  return MOBI_OK;
}
static int gop_enqueue_parse(mobi_batch *b, mobi_batch::GopSlot &S, hipEvent_t after = nullptr) {
  const int n = b->n, K = S.K;
  const size_t nv = (size_t)n * K;
  S.is_host.assign(b->on_host.begin(), b->on_host.end());
  uint32_t *blen = (uint32_t *)(S.h_stage.p + nv * 8);
  DpStaged st;
  for (size_t v = 0; v < nv; v++) {
    if (S.is_host[v % n]) { blen[v] = MOBI_DP_SKIP; continue; }
    blen[v] = S.lens[v];
    st.n_dev++;
    st.n_iframes += S.lens[v] >= 2 && (S.h_stage.p[S.hdr_bytes + S.boff[v] + 1] & 0x80) != 0;
  }
  S.lockstep = ls_decide(b, st);
  const size_t cap_words = S.cap_words; // (gop_reserve, when the group was begun)
  hipStream_t ps = b->stream_p;
  HIP_TRY(hipStreamWaitEvent(ps, S.ev_up, 0));
  // `after`: the reconstruction steps just enqueued for the group in front go FIRST.  A full parse workgroup takes a CU's whole LDS (36 lanes x
  // 8 waves), the octet kernel's workgroups the rest: left to the dispatcher, the parse got in first, the reconstruction the caller is waiting
  // for sat behind it, and by the time mobi_batch_gop_finish returned the GPU had nothing left to do while the host gathered the next group
  // (24576 clips x 6: 147 ms per group against 62 + 50 of kernels).  In this order the host's turn overlaps the parse.
  if (after) HIP_TRY(hipStreamWaitEvent(ps, after, 0));
  HIP_TRY(hipMemcpyAsync(S.d_bits.p, S.h_stage.p, S.hdr_bytes, hipMemcpyHostToDevice, ps)); // offsets and lengths (with the host parser's clips marked)
  MobiGopArgs G;
  memset(&G, 0, sizeof(G));
  MobiDevParseArgs &pa = G.P;
  pa.bits = S.d_bits.p + S.hdr_bytes;
  pa.bit_off = (const uint64_t *)S.d_bits.p;
  pa.bit_len = (const uint32_t *)(S.d_bits.p + nv * 8);
  pa.tables = b->d_ptables;
  pa.state_in = (const MobiDevState *)S.d_sin.p; pa.state_out = (MobiDevState *)S.d_sout.p;
  pa.tail_in = nullptr; pa.tail_out = (MobiDevTail *)S.d_tails.p;
  pa.scale = b->d_scale;
  pa.state_ls = (MobiDevState *)S.d_sls.p;
  pa.lockstep = S.lockstep ? 2 : 0;
  pa.pay_local = 1;
  pa.clip_mod = n;
  pa.skip_tail = 1;
  pa.desc = (MbDesc *)S.d_desc.p;
  pa.payload = (uint32_t *)S.d_pay.p;
  pa.items = (uint32_t *)S.d_items.p;
  pa.res = (MobiDevResult *)S.d_res.p;
  pa.pay_cap = (uint32_t)cap_words;
  pa.n_clips = (int)nv; pa.version = b->version;
  pa.width = b->g.width; pa.height = b->g.height; pa.stride = b->g.stride; pa.lg = b->g.lg; pa.mbw = b->g.mbw; pa.mbh = b->g.mbh;
  const int in = b->ps_cur, out = (b->ps_cur + 1) % 3;
  G.ring_in = b->d_pstate[in]; G.ring_out = b->d_pstate[out];
  G.rtail_in = b->d_ptail[in]; G.rtail_out = b->d_ptail[out];
  G.n = n; G.K = K;
  if (b->ktiming && !b->ev_p0) { (void)hipEventCreate(&b->ev_p0); (void)hipEventCreate(&b->ev_p1); }
  const bool ptime = b->ktiming && b->ev_p0 && b->ev_p1;
  if (ptime) (void)hipEventRecord(b->ev_p0, ps);
  if (mobi_launch_gop_prepare(&G, ps) != 0) return MOBI_E_DEVICE;
  if (mobi_launch_parse(&pa, ps) != 0) return MOBI_E_DEVICE;
  if (mobi_launch_gop_chain(&G, ps) != 0) return MOBI_E_DEVICE;
  if (ptime) (void)hipEventRecord(b->ev_p1, ps);
  HIP_TRY(hipMemcpyAsync(S.h_res.p, S.d_res.p, nv * sizeof(MobiDevResult), hipMemcpyDeviceToHost, ps));
  HIP_TRY(hipEventRecord(S.ev_parsed, ps));
  b->ps_cur = out;
  S.ring_in = in;
  S.parse_enqueued = true;
  return MOBI_OK;
}

int mobi_batch_gop_begin(mobi_batch *b, int n_frames, const uint8_t *const *data, const size_t *len, const int32_t *offsets) {
  if (!b || !data || !len || !offsets || n_frames < 1 || n_frames > MOBI_GOP_PARSE_MAX) return MOBI_E_ARG;
  HIP_TRY(hipSetDevice(b->device));
  const int n = b->n, K = n_frames, n_mbs = b->g.mbw * b->g.mbh;
  const size_t nv = (size_t)n * K;
  if (b->frames_started == 0 && b->async_seq == 0 && b->gop_count == 0) { // a batch fed by groups parses on the GPU from its first frame
    if (b->parse_auto) { b->parse_mode = 1; b->parse_auto = false; }
    if (b->parse_mode == 2) hybrid_share(b);
  }
  if (b->poisoned) return MOBI_E_DEVICE;
  if (b->parse_mode == 0) return MOBI_E_ARG; // the decoder state of this batch lives in the host parsers (mobi_batch_decode_gop serves those too)
  if (b->version != MOBI_VERSION_MODSDS && b->version != MOBI_VERSION_MOFLEX3DS) return MOBI_E_VERSION;
  if (b->g.mbw > 64 || b->async_count || b->gop_count >= 2 || nv >= ((size_t)1 << 22)) return MOBI_E_ARG;
  if (int e = dp_init(b)) return e;
  if (b->hybrid_host && b->frames_started == 0 && b->gop_count == 0)
    for (int i = n - b->hybrid_host; i < n; i++) b->on_host[i] = b->host_share[i] = 1;
  mobi_batch::GopSlot &S = b->gslot[(b->gop_head + b->gop_count) & 1];
  if (!S.ev_up) HIP_TRY(hipEventCreateWithFlags(&S.ev_up, hipEventDisableTiming));
  if (!S.ev_parsed) HIP_TRY(hipEventCreateWithFlags(&S.ev_parsed, hipEventDisableTiming));
  if (!b->stream2) HIP_TRY(hipStreamCreateWithFlags(&b->stream2, hipStreamNonBlocking));
  if (!b->stream_p) HIP_TRY(hipStreamCreateWithFlags(&b->stream_p, hipStreamNonBlocking));
  const auto t_stage0 = std::chrono::steady_clock::now();
  // the staged image: [bit_off u64 x nv][bit_len u32 x nv][bits: every frame 8-byte aligned, 32 zero bytes behind it]
  constexpr size_t kBitPad = 32;
  S.K = K;
  S.boff.resize(nv); S.lens.resize(nv);
  S.offs.assign(offsets, offsets + nv);
  size_t pos = 0, max_len = 0;
  const size_t frame_bound = (size_t)n_mbs * 4096 + 64; // (dp_stage: bytes beyond cannot influence the parse of one frame)
  for (size_t v = 0; v < nv; v++) {
    const int64_t o = offsets[v];
    size_t l = (data[v] && o >= 0 && (uint64_t)o < len[v]) ? len[v] - (size_t)o : 0;
    l = std::min(l, frame_bound);
    max_len = std::max(max_len, l);
    S.boff[v] = pos;
    S.lens[v] = (uint32_t)l;
    pos += align_up(l + kBitPad, 8);
  }
  pos += 64;
  const size_t hdr_bytes = align_up(nv * 12, 16), need = hdr_bytes + pos;
  if (need > S.h_stage.cap)
    if (int e = S.h_stage.reserve(need + need / 4)) return e;
  if (need > S.d_bits.cap)
    if (int e = S.d_bits.reserve(need + need / 4)) return e;
  uint8_t *hs = S.h_stage.p;
  memcpy(hs, S.boff.data(), nv * 8);
  memset(hs + hdr_bytes + pos - 64, 0, 64);
  S.hdr_bytes = hdr_bytes; S.bytes = need; S.max_len = max_len;
  auto gather = [&](size_t v) {
    uint8_t *dst = hs + hdr_bytes + S.boff[v];
    const size_t l = S.lens[v];
    if (l) memcpy(dst, data[v] + offsets[v], l);
    memset(dst + l, 0, align_up(l + kBitPad, 8) - l);
  };
  // gathered in chunks by the pool while one of its threads hands the chunk before to the copy engine (dp_stage)
  {
    const int chunks = nv >= 2048 ? 8 : 1;
    std::atomic<int> up_err{0};
    auto start_of = [&](size_t v) { return v < nv ? hdr_bytes + (size_t)S.boff[v] : need; };
    for (int k = 0; k <= chunks; k++) {
      const size_t c0 = k < chunks ? nv * k / chunks : nv, c1 = k < chunks ? nv * (k + 1) / chunks : nv;
      const size_t u0 = k >= 1 ? nv * (k - 1) / chunks : 0, u1 = k >= 1 ? nv * k / chunks : 0;
      const int n_up = u1 > u0 ? 1 : 0;
      if (n_up + (c1 - c0) == 0) continue;
      constexpr size_t kRun = 64; // frames per task: one per frame had the pool's threads queue at its counter (122 880 frames of 256x192: 8 ms)
      b->pool->run(n_up + (int)((c1 - c0 + kRun - 1) / kRun), [&](int j) {
        if (j < n_up) {
          const size_t a = start_of(u0), e = start_of(u1);
          if (e > a && (hipSetDevice(b->device) != hipSuccess || hipMemcpyAsync(S.d_bits.p + a, hs + a, e - a, hipMemcpyHostToDevice, b->stream2) != hipSuccess)) up_err = 1;
          return;
        }
        for (size_t v = c0 + (size_t)(j - n_up) * kRun, e = std::min(c1, v + kRun); v < e; v++) gather(v);
      });
    }
    if (up_err) { (void)hipStreamSynchronize(b->stream2); return MOBI_E_DEVICE; }
  }
  HIP_TRY(hipEventRecord(S.ev_up, b->stream2));
  b->last_stage_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_stage0).count();
  S.parse_enqueued = false;
  S.resolved = S.returned = false;
  S.done = 0;
  if (int e = gop_reserve(b, S)) { (void)hipStreamSynchronize(b->stream2); return e; } // (does not fit: refused, nothing of it is in flight)
  b->gop_count++;
  if (b->gop_count == 1) // nothing in front: the parse may start at once (else mobi_batch_gop_finish of the group in front enqueues it, once it knows whose clips are whose)
    if (int e = gop_enqueue_parse(b, S)) { (void)hipStreamSynchronize(b->stream2); (void)hipStreamSynchronize(b->stream_p); b->gop_count--; return e; }
  return MOBI_OK;
}

int mobi_batch_gop_finish(mobi_batch *b, int32_t *offsets_out, int *rc) {
  if (!b || !rc) return MOBI_E_ARG;
  HIP_TRY(hipSetDevice(b->device));
  if (b->poisoned) return MOBI_E_DEVICE;
  if (b->gop_count == 0) return MOBI_E_ARG;
  mobi_batch::GopSlot &S = b->gslot[b->gop_head & 1];
  const int n = b->n, K = S.K, n_mbs = b->g.mbw * b->g.mbh;
  const size_t nv = (size_t)n * K;
  // whatever fails from here on leaves rings, parsers and state ring out of step: the batch is drained and refuses further work
  struct Poison {
    mobi_batch *b; bool armed = true;
    ~Poison() {
      if (!armed) return;
      if (b->stream2) (void)hipStreamSynchronize(b->stream2);
      if (b->stream_p) (void)hipStreamSynchronize(b->stream_p);
      (void)hipStreamSynchronize(b->stream);
      b->poisoned = true;
    }
  } poison{b};
  if (!S.parse_enqueued)
    if (int e = gop_enqueue_parse(b, S)) return e;
  using clk = std::chrono::steady_clock;
  const auto q0 = clk::now();
  auto ms_since = [](clk::time_point x) { return std::chrono::duration<float, std::milli>(clk::now() - x).count(); };
  std::vector<int> &host_from = S.host_from, &hslot = S.hslot, &hrc = S.hrc, &all_host = S.all_host;
  std::vector<int32_t> &hoff = S.hoff;
  std::vector<uint32_t> &hq = S.hq, &hy = S.hy;
  std::vector<uint8_t> &hready = S.hready;
  const size_t desc_b = (size_t)n_mbs * sizeof(MbDesc), item_b = (size_t)n_mbs * 4;
  if (!S.resolved) { // the group's first finish: whose frames are whose, for all K of them
    // 1. the host parser's clips, all K frames of each, while the GPU parses the others
    host_from.assign(n, K); // first frame of the group that is the host parser's
    hslot.assign(n, -1);    // its place in gop_frames
    hrc.assign(nv, MOBI_OK);
    hoff.assign(nv, 0);
    hq.assign(nv, 0); hy.assign(nv, 0); // Quantizer / YuvFormat behind a host-parsed frame
    hready.assign(nv, 0);               // a host-parsed frame the device parsers would have finished too (MobiStreamParser::device_ready)
    std::vector<int> host_clips;
    for (int c = 0; c < n; c++)
      if (S.is_host[c]) { host_from[c] = 0; hslot[c] = (int)host_clips.size(); host_clips.push_back(c); }
    auto host_parse = [&](const std::vector<int> &cl) {
      b->pool->run((int)cl.size(), [&](int j) {
        const int c = cl[j];
        for (int k = host_from[c]; k < K; k++) {
          const size_t v = (size_t)k * n + c;
          int32_t off = 0;
          hrc[v] = b->parsers[c]->parse_frame(S.h_stage.p + S.hdr_bytes + S.boff[v], S.lens[v], &off, b->gop_frames[(size_t)hslot[c] * K + k]);
          hoff[v] = S.offs[v] + off;
          hready[v] = hrc[v] == MOBI_OK && b->parsers[c]->device_ready();
          hq[v] = b->parsers[c]->quantizer();
          hy[v] = b->parsers[c]->yuv_format();
        }
      });
    };
    if (b->gop_frames.size() < host_clips.size() * K) b->gop_frames.resize(host_clips.size() * K);
    host_parse(host_clips);
    // 2. what the device parsers finished
    HIP_TRY(hipEventSynchronize(S.ev_parsed));
    b->phase_ms[0] = ms_since(q0);
    { const bool ptime = b->ktiming && b->ev_p0 && b->ev_p1; float ms = 0; if (ptime && hipEventElapsedTime(&ms, b->ev_p0, b->ev_p1) == hipSuccess) b->last_parse_ms = ms; }
    const MobiDevResult *res = (const MobiDevResult *)S.h_res.p;
    std::vector<int> fb;
    for (int c = 0; c < n; c++) {
      if (S.is_host[c]) continue;
      for (int k = 0; k < K; k++)
        if (res[(size_t)k * n + c].rc != MOBI_OK) { host_from[c] = k; fb.push_back(c); break; }
    }
    if (!fb.empty()) { // hand-overs: the state the first unfinished frame started from (mobi_gop_chain left the true one in its start slot) and the tail before it
      const size_t rec = sizeof(MobiDevState) + sizeof(MobiDevTail);
      if (int e = S.h_seed.reserve(fb.size() * rec)) return e;
      for (size_t j = 0; j < fb.size(); j++) {
        const int c = fb[j], k = host_from[c];
        const size_t v = (size_t)k * n + c;
        HIP_TRY(hipMemcpyAsync(S.h_seed.p + j * rec, (const MobiDevState *)S.d_sin.p + v, sizeof(MobiDevState), hipMemcpyDeviceToHost, b->stream_p));
        const MobiDevTail *t = k == 0 ? b->d_ptail[S.ring_in] + c : (const MobiDevTail *)S.d_tails.p + (v - n);
        HIP_TRY(hipMemcpyAsync(S.h_seed.p + j * rec + sizeof(MobiDevState), t, sizeof(MobiDevTail), hipMemcpyDeviceToHost, b->stream_p));
      }
      HIP_TRY(hipStreamSynchronize(b->stream_p));
      const size_t base = host_clips.size();
      if (b->gop_frames.size() < (base + fb.size()) * K) b->gop_frames.resize((base + fb.size()) * K);
      for (size_t j = 0; j < fb.size(); j++) {
        const int c = fb[j];
        hslot[c] = (int)(base + j);
        b->parsers[c]->import_state(*(const MobiDevState *)(S.h_seed.p + j * rec), *(const MobiDevTail *)(S.h_seed.p + j * rec + sizeof(MobiDevState)));
      }
      host_parse(fb);
      for (int c : fb) {
        b->on_host[c] = 1;
        b->clean_run[c] = 0;
        b->clean_need[c] = (uint16_t)std::min(256, 2 * (int)b->clean_need[c]);
      }
      b->fallbacks += fb.size();
    }
    b->phase_ms[1] = ms_since(q0);
    // 3. the host parser's command lists over the rows the parse kernels left, frame by frame
    HIP_TRY(hipStreamWaitEvent(b->stream, S.ev_parsed, 0));
    all_host = host_clips;
    all_host.insert(all_host.end(), fb.begin(), fb.end());
    std::sort(all_host.begin(), all_host.end());
    for (int k = 0; k < K && !all_host.empty(); k++) {
      std::vector<int> cl;
      std::vector<const ParsedFrame *> fr;
      std::vector<int> rcs;
      for (int c : all_host)
        if (host_from[c] <= k) {
          const size_t v = (size_t)k * n + c;
          cl.push_back(c);
          rcs.push_back(hrc[v]);
          fr.push_back(hrc[v] == MOBI_OK ? &b->gop_frames[(size_t)hslot[c] * K + k] : nullptr);
        }
      const size_t kn = (size_t)k * n;
      const DpRows rows{S.d_desc.p + kn * desc_b, S.d_pay.p + kn * S.cap_words * 4, S.d_items.p + kn * item_b, (MobiDevResult *)S.d_res.p + kn, S.cap_words};
      if (int e = dp_override(b, cl, nullptr, S.h_over[k], rows, b->stream, fr.data(), rcs.data())) return e;
    }
    HIP_TRY(hipMemsetAsync(S.d_fault.p, 0, nv * sizeof(int), b->stream));
    // 3b. the intra macroblocks of every frame of the group as launch items in wavefront order, built on the device behind the overrides (mobi_gop.h)
    // (the wavefront order holds where the halo's linear addresses do not wrap: in a picture as wide as its stride -- 256, 512, 1024 -- the
    // first macroblock of a row reads the LAST one of the row above, MD.cs:212-217 with Stride == Width; those keep the raster-order launch)
    S.sorted = b->g.width < b->g.stride;
#if defined(MOBI_PROFILING)
    if (const char *e = getenv("MOBI_GOP_INTRA_SORT")) S.sorted = S.sorted && atoi(e) != 0; // (A/B: 0 = the raster-order launch, mobi_recon_intra_cl)
#endif
    if (S.sorted) {
      MobiGopSortArgs A;
      memset(&A, 0, sizeof(A));
      uint64_t off = 0;
      for (int k = 0; k < K; k++) {
        uint64_t sum = 0;
        for (int c = 0; c < n; c++) {
          const size_t v = (size_t)k * n + c;
          sum += k >= host_from[c] ? (hrc[v] == MOBI_OK ? b->gop_frames[(size_t)hslot[c] * K + k].hdr.n_intra : 0u) : res[v].n_intra;
        }
        S.sorted_items[k] = sum ? (uint32_t)(align_up(sum, 4) + 3 * MOBI_SORT_LEVELS + 4) & ~3u : 0u; // (every wavefront starts on a wave of four: at most three rows of padding each)
        S.sorted_off[k] = off;
        A.sorted_off[k] = off;
        A.sorted_cap[k] = S.sorted_items[k];
        off += S.sorted_items[k];
      }
      if (off) {
        const size_t hist_b = (size_t)K * MOBI_SORT_LEVELS * 4;
        if (int e = S.d_sorted.reserve(off * 16)) return e;
        if (int e = S.d_hist.reserve(3 * hist_b)) return e;
        HIP_TRY(hipMemsetAsync(S.d_hist.p, 0, 2 * hist_b, b->stream));
        A.desc = (const MbDesc *)S.d_desc.p;
        A.items = (const uint32_t *)S.d_items.p;
        A.res = (const MobiDevResult *)S.d_res.p;
        A.hist = (uint32_t *)S.d_hist.p;
        A.cursor = (uint32_t *)(S.d_hist.p + hist_b);
        A.start = (uint32_t *)(S.d_hist.p + 2 * hist_b);
        A.sorted = (uint32_t *)S.d_sorted.p;
        A.n = n; A.K = K; A.n_mbs = n_mbs; A.mbw = b->g.mbw;
        if (mobi_launch_gop_sort(&A, b->stream) != 0) return MOBI_E_DEVICE;
      }
    }
    S.resolved = true;
    S.done = 0;
  }
  const MobiDevResult *res = (const MobiDevResult *)S.h_res.p;
  // 4. the reconstruction steps of this part -- at most six: the ring holds six pictures -- from consecutive command lists
  const int k0 = S.done, k1 = std::min(K, k0 + 6);
  const bool last = k1 == K;
  for (int k = k0; k < k1; k++) {
    const size_t kn = (size_t)k * n;
    uint32_t Kint = 0;
    for (int c = 0; c < n; c++) {
      const size_t v = kn + c;
      if (k >= host_from[c]) { if (hrc[v] == MOBI_OK) Kint = std::max(Kint, b->gop_frames[(size_t)hslot[c] * K + k].hdr.n_intra); }
      else Kint = std::max(Kint, res[v].n_intra);
    }
    b->ring_base = (b->ring_base + 1) % 6; // Y[i] = Y[i-1]; Y[0] = new (MD.cs:102-108) -- even if the parse threw
    b->step_tag = b->step_tag + 1 ? b->step_tag + 1 : 1;
    b->argb_all_valid = false;
    b->frames_started++;
    MobiReconArgs a = b->args(S.d_desc.p + kn * desc_b, S.d_pay.p + kn * S.cap_words * 4);
    a.pay_clip_words = (uint32_t)S.cap_words;
    a.fault = (int *)S.d_fault.p + kn;
    mobi_batch::EvPair ep{nullptr, nullptr, 0};
    if (b->ktiming) { ep.a = b->get_event(); ep.b = b->get_event(); (void)hipEventRecord(ep.a, b->stream); }
    if (mobi_launch_inter(&a, b->stream) != 0) return MOBI_E_DEVICE;
    if (b->ktiming) { (void)hipEventRecord(ep.b, b->stream); b->evs.push_back(ep); }
    const MobiDevResult *d_res_k = (const MobiDevResult *)S.d_res.p + kn;
    if (S.sorted) {
      if (S.sorted_items[k] && mobi_launch_intra(&a, (const uint32_t *)S.d_sorted.p + S.sorted_off[k] * 4, (int)S.sorted_items[k], b->stream) != 0) return MOBI_E_DEVICE;
    } else if (Kint && mobi_launch_intra_cl(&a, (const uint32_t *)(S.d_items.p + kn * item_b), &d_res_k->n_intra, (int)(sizeof(MobiDevResult) / 4), (int)Kint, 0, b->stream) != 0)
      return MOBI_E_DEVICE;
  }
  b->pay_clip_words = (uint32_t)S.cap_words;
  HIP_TRY(hipMemcpyAsync(S.h_fault.p + (size_t)k0 * n * sizeof(int), S.d_fault.p + (size_t)k0 * n * sizeof(int), (size_t)(k1 - k0) * n * sizeof(int), hipMemcpyDeviceToHost, b->stream));
  b->phase_ms[2] = ms_since(q0);
  // (what the next parse needs to know -- whose clips are whose -- is settled once the group is resolved: the clips that go back are sent with the first part)
  if (!S.returned) {
    S.returned = true;
    // 5. clips that go back to the device parsers (dp_return's rule, counted in frames): their state into the entry the next parse reads
    {
      std::vector<int> back;
      for (int c : all_host) {
        if (b->host_share[c]) continue;
        int run = b->clean_run[c]; // consecutive frames the device parsers would have finished too
        for (int k = host_from[c]; k < K; k++) run = hready[(size_t)k * n + c] ? run + 1 : 0;
        b->clean_run[c] = (uint16_t)std::min(60000, run);
        if (run >= b->clean_need[c]) back.push_back(c);
      }
      if (int e = dp_return_list(b, back, b->ps_cur, S.h_ret, b->stream_p)) return e;
    }
  }
  // 6. the group begun behind this one: its parse goes out behind the reconstruction steps of this group's LAST part.  Parse and reconstruction
  // do not share the GPU (a full parse workgroup takes a CU's whole LDS), so the parse belongs where only the host is busy: a part's steps
  // are waited for before the call returns, the caller's next move behind the last part is the gather and upload of the group after next
  // (mobi_batch_gop_begin: tens of ms at 131 072 frames) -- with the parse in the queue the GPU works through that; with the parse sent out
  // with the FIRST part (r06 until its last hours) it ran while the host sat in the second part's wait, and the GPU idled through the gather
  // (4096 clips x 32 frames: 3.4 -> 2.9 - 3.2 ms per frame step; groups of one part are the same either way).
  bool parse_now = last;
#if defined(MOBI_PROFILING)
  static const int parse_first = getenv("MOBI_GOP_PARSE_FIRST") ? atoi(getenv("MOBI_GOP_PARSE_FIRST")) : 0; // (A/B: with the group's first part)
  if (parse_first) parse_now = true;
#endif
  if (b->gop_count == 2 && parse_now) {
    mobi_batch::GopSlot &N = b->gslot[(b->gop_head + 1) & 1];
    if (!N.parse_enqueued) {
      if (!S.ev_recon) HIP_TRY(hipEventCreateWithFlags(&S.ev_recon, hipEventDisableTiming));
      HIP_TRY(hipEventRecord(S.ev_recon, b->stream));
      if (int e = gop_enqueue_parse(b, N, S.ev_recon)) return e;
    }
  }
  HIP_TRY(hipStreamSynchronize(b->stream));
  b->phase_ms[3] = ms_since(q0);
  b->drain_events();
  poison.armed = false;
  const int *fault = (const int *)S.h_fault.p;
  if (k0 == 0) b->ls_finished = S.lockstep ? 0 : -1;
  b->lockstep = S.lockstep;
  for (int k = k0; k < k1; k++)
    for (int c = 0; c < n; c++) {
      const size_t v = (size_t)k * n + c, o = (size_t)(k - k0) * n + c; // (the part's results count from its first frame)
      if (k >= host_from[c]) {
        rc[o] = hrc[v];
        if (offsets_out) offsets_out[o] = hoff[v];
      } else {
        if (S.lockstep && res[v].pad == MOBI_LS_MAGIC) b->ls_finished++;
        rc[o] = res[v].rc;
        if (offsets_out) offsets_out[o] = S.offs[v] + (int32_t)res[v].consumed;
      }
      if (rc[o] == MOBI_OK && fault[v]) rc[o] = (fault[v] & 2) ? MOBI_E_DEVICE : MOBI_E_CLAMP;
      if (k == k1 - 1) {
        b->dev_quant[c] = k >= host_from[c] ? hq[v] : res[v].quant;
        b->dev_yuvfmt[c] = k >= host_from[c] ? hy[v] : res[v].yuvfmt;
      }
    }
  S.done = k1;
  if (last) {
    b->gop_head ^= 1;
    b->gop_count--;
    S.parse_enqueued = false;
    S.resolved = S.returned = false;
  }
  return MOBI_OK;
}
// frames of the OLDEST group begun that mobi_batch_gop_finish has not reported yet (0: no group is begun); the next finish reports min(6, that)
int mobi_batch_gop_frames_pending(const mobi_batch *b) {
  if (!b || b->gop_count == 0) return 0;
  const mobi_batch::GopSlot &S = b->gslot[b->gop_head & 1];
  return S.K - (S.resolved ? S.done : 0);
}
int mobi_batch_gop_in_flight(const mobi_batch *b) { return b ? b->gop_count : 0; }

int mobi_batch_decode_gop(mobi_batch *b, int n_frames, const uint8_t *const *data, const size_t *len, int32_t *offsets, int *rc) {
  if (!b || !data || !len || !offsets || !rc || n_frames < 1 || n_frames > MOBI_GOP_MAX) return MOBI_E_ARG;
  if (b->poisoned) return MOBI_E_DEVICE;
  if (b->async_count || b->gop_count) return MOBI_E_ARG;
  const int n = b->n;
  if (b->parse_auto && b->frames_started == 0) { // as mobi_batch_decode chooses, with the lanes a group offers counted: n_clips * n_frames
    b->parse_mode = (size_t)n * n_frames >= (size_t)std::max(640, 20 * (b->pool->size() + 1));
    b->parse_auto = false;
    if (b->parse_mode == 2) hybrid_share(b);
  }
  if (b->parse_mode == 0 || (size_t)n * n_frames >= ((size_t)1 << 22) || (b->version != MOBI_VERSION_MODSDS && b->version != MOBI_VERSION_MOFLEX3DS)) {
    for (int k = 0; k < n_frames; k++) // the host parser's batch (or a version no parser knows): K calls, the same results
      if (int e = mobi_batch_decode(b, data + (size_t)k * n, len + (size_t)k * n, offsets + (size_t)k * n, rc + (size_t)k * n)) return e;
    return MOBI_OK;
  }
  if (int e = mobi_batch_gop_begin(b, n_frames, data, len, offsets)) return e;
  return mobi_batch_gop_finish(b, offsets, rc);
}

int mobi_batch_in_flight(const mobi_batch *b) { return b ? b->async_count : 0; }
int mobi_batch_host_clips(const mobi_batch *b) {
  if (!b) return 0;
  if (!b->parse_mode) return b->n;
  int k = 0;
  for (uint8_t h : b->on_host) k += h;
  return k;
}
int mobi_batch_lockstep_finished(const mobi_batch *b) { return b ? b->ls_finished : -1; }

int mobi_batch_set_parse_mode(mobi_batch *b, int device_parse) {
  if (!b || b->frames_started != 0) return MOBI_E_ARG; // the decoder state lives either in the host parsers or in HBM, not both
  b->parse_mode = device_parse == 2 ? 2 : device_parse != 0;
  b->ls_policy = device_parse == 3 ? 1 : 0;
  b->parse_auto = false;
  return MOBI_OK;
}

int mobi_batch_decode(mobi_batch *b, const uint8_t *const *data, const size_t *len, int32_t *offsets, int *rc) {
  if (!b || !data || !len || !offsets || !rc) return MOBI_E_ARG;
  if (b->poisoned) return MOBI_E_DEVICE;
  HIP_TRY(hipSetDevice(b->device));
  struct CallTimer { // wall time of this call, for the end-to-end measurements (tools/exp_dparse.py)
    mobi_batch *b;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    ~CallTimer() { b->last_decode_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
  } call_timer{b};
  if (b->gop_count) return MOBI_E_ARG; // (groups in flight: mobi_batch_gop_finish first)
  if (b->parse_auto && b->parse_mode && b->frames_started == 0) {
    // Chosen by batch size only: a caller that hands over whole files as Data (MOC5 style, Form1.cs:292-302) would make the
    // device path upload up to 4 KB per macroblock per clip and frame (the frame length is unknown before the parse).  Packets
    // of a frame or a few are what the device path is for; anything much larger stays on the host parser.
    const size_t packet_like = (size_t)b->g.mbw * b->g.mbh * 256 + 65536;
    for (int i = 0; i < b->n; i++)
      if (data[i] && offsets[i] >= 0 && (uint64_t)offsets[i] < len[i] && len[i] - (size_t)offsets[i] > packet_like) { b->parse_mode = 0; break; }
  }
  b->parse_auto = false; // the decoder state lives on one side from the first frame on
  if (b->parse_mode == 2 && b->frames_started == 0) { // hybrid: a fixed share of the clips stays with the host parsers (their state lives there)
    hybrid_share(b);
  }
  if (b->parse_mode) return decode_device_parse(b, data, len, offsets, rc);
  const int n = b->n;
  // 1. host: serial VLC parse of one frame per clip -> command lists.  The clips are taken in chunks, three stages deep: in one round of
  // the pool chunk k is parsed, chunk k - 1 is written into the staging buffer, and chunk k - 2 is handed to the copy engine (a step of
  // 1024 clips of 640x480 is 100 MB of commands: as long on the bus as it is in the parsers).  A chunk's place in the payload arena is the
  // sum of the chunks before it, so nothing has to wait for the whole batch; only the launch list does (levels are sorted over all clips).
  // The pipelined path needs the step to fit the buffers as they are (they grow in the classic path below, with headroom): the first frame
  // of a batch, and any frame a quarter larger than every one before it, is staged and uploaded after the parse, as all were before r04.
  std::vector<const ParsedFrame *> ok(n, nullptr);
  bool any_version_error = false;
  static const uint8_t kNoData[2] = {0, 0};
  const int n_mbs = b->g.mbw * b->g.mbh;
  const size_t desc_bytes = align_up((size_t)n * n_mbs * sizeof(MbDesc) + 8 * sizeof(MbDesc), kAlign); // slack: a wave reads up to 8 descriptors at once
  const size_t buf_cap = std::min(b->h_stage.cap, b->d_cmd.cap);
  const size_t cap_words = buf_cap > desc_bytes + kPaySlack + kAlign ? (buf_cap - desc_bytes - kPaySlack - kAlign) / 4 : 0;
  const int chunk = b->host_chunk;
  const int chunks = chunk > 0 && n >= 2 * chunk ? (n + chunk - 1) / chunk : 1;
  bool piped = chunks > 1 && cap_words > 0;
  int uploaded = 0; // clips [0, uploaded) are staged and on their way
  std::vector<size_t> base(n + 1, 0); // where each clip's payload starts in the step's arena (words)
  uint8_t *hs = b->h_stage.p;
  float parse_ms = 0;
  auto stage_range = [&](int c0, int c1) {
    for (int i = c0; i < c1; i++) step_write_clip(ok[i], base[i], n_mbs, (MbDesc *)hs + (size_t)i * n_mbs, (uint32_t *)(hs + desc_bytes));
  };
  auto stage_clips = [&](int c0, int c1) { // every clip writes its own descriptors and payload (at most 32 writers: more threads than that
    const int groups = std::min(c1 - c0, 32); // on the pinned buffer slow each other down -- measured: 34 ms -> 70-90 ms per step of 2048 clips)
    b->pool->run(groups, [&](int g) { stage_range(c0 + (int)((long)(c1 - c0) * g / groups), c0 + (int)((long)(c1 - c0) * (g + 1) / groups)); });
  };
  std::atomic<int> up_err{0};
  auto upload_clips = [&](int c0, int c1) { // (the copy calls hold their thread for as long as the bus is busy: 2.3 ms per 100 MB, measured)
    if (hipSetDevice(b->device) != hipSuccess) { up_err = 1; return; }
    const size_t d0 = (size_t)c0 * n_mbs * sizeof(MbDesc), d1 = (size_t)c1 * n_mbs * sizeof(MbDesc);
    if (hipMemcpyAsync(b->d_cmd.p + d0, hs + d0, d1 - d0, hipMemcpyHostToDevice, b->stream) != hipSuccess) up_err = 1;
    if (base[c1] > base[c0] &&
        hipMemcpyAsync(b->d_cmd.p + desc_bytes + base[c0] * 4, hs + desc_bytes + base[c0] * 4, (base[c1] - base[c0]) * 4, hipMemcpyHostToDevice, b->stream) != hipSuccess)
      up_err = 1;
  };
  // Round k of the pool: the clips of chunk k are parsed; beside them chunk k - 1 (parsed, its place in the arena known) is staged in
  // eight pieces, and chunk k - 2 (staged) is handed to the copy engine by one thread.
  int staged = 0;        // clips [0, staged) are in the staging buffer
  for (int k = 0; k < chunks + 2; k++) {
    const int c0 = k < chunks ? (int)((long)n * k / chunks) : n, c1 = k < chunks ? (int)((long)n * (k + 1) / chunks) : n;
    const int s0 = staged, s1 = piped && k >= 1 ? (int)((long)n * std::min(k, chunks) / chunks) : staged;         // to stage now: parsed chunks not staged yet
    const int u0 = uploaded, u1 = piped ? staged : uploaded;                                                       // to upload now: staged, not uploaded
    const int n_parse = c1 - c0, n_stage = s1 > s0 ? std::min(8, s1 - s0) : 0, n_up = u1 > u0 ? 1 : 0;
    if (n_parse + n_stage + n_up == 0) continue;
    const auto t0 = std::chrono::steady_clock::now();
    b->pool->run(n_up + n_stage + n_parse, [&](int j) {
      if (j < n_up) { upload_clips(u0, u1); return; }
      j -= n_up;
      if (j < n_stage) { stage_range(s0 + (int)((long)(s1 - s0) * j / n_stage), s0 + (int)((long)(s1 - s0) * (j + 1) / n_stage)); return; }
      const int i = c0 + (j - n_stage);
      rc[i] = b->parsers[i]->parse_frame(data[i] ? data[i] : kNoData, data[i] ? len[i] : 0, &offsets[i], b->cur[i]); // Data == null: nothing readable, as Data.Length == 0
    });
    parse_ms += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (up_err) piped = false; // (reported below, once the ring has turned and every clip's rc can say so: the parsers have consumed their frames)
    uploaded = u1;
    staged = s1;
    for (int i = c0; i < c1; i++) {
      if (rc[i] == MOBI_OK) ok[i] = &b->cur[i];
      if (rc[i] == MOBI_E_VERSION) any_version_error = true;
      base[i + 1] = base[i] + (ok[i] ? ok[i]->payload.size() : 0);
    }
    if (any_version_error || base[c1] > cap_words) piped = false; // (what is staged already still goes up if it was asked to; the rest waits for the classic path)
  }
  const auto tp1 = std::chrono::steady_clock::now();
  b->last_hostparse_ms = parse_ms;
  if (any_version_error) { // DecodeFrame() returns before touching the ring (MD.cs:56-61)
    if (uploaded) HIP_TRY(hipStreamSynchronize(b->stream)); // (nothing may still be reading the staging buffer when the next call fills it)
    return MOBI_OK;
  }
  b->ring_base = (b->ring_base + 1) % 6; // Y[i] = Y[i-1]; Y[0] = new (MD.cs:102-108) -- even if the parse threw
  b->step_tag = b->step_tag + 1 ? b->step_tag + 1 : 1;
  b->argb_all_valid = false;
  b->frames_started++;
  // From here on the parsers have consumed the frame and the ring has turned (the two must stay in step: a parser's reference
  // bookkeeping counts frames).  If the call itself fails below, no clip may report MOBI_OK for a frame that was never reconstructed.
  FailAll fail_all{rc, n, b->stream};
  if (up_err) return MOBI_E_DEVICE;
  if (base[n] + kPaySlack / 4 >= ((uint64_t)1 << 32)) return MOBI_E_ARG; // MbDesc.payload_off is a 32-bit word offset into the step's arena
  LevelPlan plan;
  plan.build(ok, b->g.mbw);
  const auto tp2 = std::chrono::steady_clock::now();
  // 2. what is not on its way yet: [desc table][payload arena] (all of it, if the step did not go chunk by chunk), and the items
  const size_t pay_bytes = align_up(base[n] * 4 + kPaySlack, kAlign);
  const size_t item_bytes = align_up(plan.items.size() * 4 + 4, kAlign);
  if (int e = b->h_items.reserve(item_bytes)) return e;
  if (int e = b->d_items.reserve(item_bytes)) return e;
  const auto t_stage0 = std::chrono::steady_clock::now();
  if (uploaded < n) {
    if (uploaded) HIP_TRY(hipStreamSynchronize(b->stream)); // the buffers may move: nothing of the chunks that did go may be in flight
    const size_t want = desc_bytes + pay_bytes + pay_bytes / 4;  // (headroom: the next steps of this size go chunk by chunk)
    if (int e = b->h_stage.reserve(want)) return e;
    if (int e = b->d_cmd.reserve(want)) return e;
    hs = b->h_stage.p;
    stage_clips(0, n);
    HIP_TRY(hipMemcpyAsync(b->d_cmd.p, hs, desc_bytes + pay_bytes, hipMemcpyHostToDevice, b->stream));
  }
  if (!plan.items.empty()) {
    memcpy(b->h_items.p, plan.items.data(), plan.items.size() * 4);
    HIP_TRY(hipMemcpyAsync(b->d_items.p, b->h_items.p, plan.items.size() * 4, hipMemcpyHostToDevice, b->stream));
  }
  b->last_stage_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_stage0).count(); // (what was not staged under the parse)
  const auto tp3 = std::chrono::steady_clock::now();
  // 3. device: reconstruction
  MobiReconArgs a = b->args(b->d_cmd.p, b->d_cmd.p + desc_bytes);
  if (int e = b->launch_plan(a, plan, (const uint32_t *)b->d_items.p)) return e;
  HIP_TRY(hipMemcpyAsync(b->h_fault.data(), b->d_fault, sizeof(int) * n, hipMemcpyDeviceToHost, b->stream));
  HIP_TRY(hipMemsetAsync(b->d_fault, 0, sizeof(int) * n, b->stream));
  const auto tp4 = std::chrono::steady_clock::now();
  HIP_TRY(hipStreamSynchronize(b->stream));
  if (b->last_fused) { // a one-launch step whose intra fours gave up waiting (fault bit 2: a dispatch order this library has never seen, see
    bool gave_up = false; // mobi_recon_step): the same step again as two launches, which need no order -- the step only writes ring slot 0
    for (int i = 0; i < n; i++) gave_up = gave_up || (b->h_fault[i] & 2);
    if (gave_up) {
      if (int e = b->launch_plan(a, plan, (const uint32_t *)b->d_items.p, -1, false)) return e;
      HIP_TRY(hipMemcpyAsync(b->h_fault.data(), b->d_fault, sizeof(int) * n, hipMemcpyDeviceToHost, b->stream));
      HIP_TRY(hipMemsetAsync(b->d_fault, 0, sizeof(int) * n, b->stream));
      HIP_TRY(hipStreamSynchronize(b->stream));
    }
  }
  {
    const auto tp5 = std::chrono::steady_clock::now();
    auto ms = [](std::chrono::steady_clock::time_point x, std::chrono::steady_clock::time_point y) { return std::chrono::duration<float, std::milli>(y - x).count(); };
    b->phase_ms[0] = ms(call_timer.t0, tp1); b->phase_ms[1] = ms(tp1, tp2); b->phase_ms[2] = ms(tp2, tp3); b->phase_ms[3] = ms(tp3, tp4); b->phase_ms[4] = ms(tp4, tp5);
  }
  fail_all.armed = false;
  b->drain_events();
  for (int i = 0; i < n; i++)
    if (rc[i] == MOBI_OK && b->h_fault[i]) rc[i] = (b->h_fault[i] & 2) ? MOBI_E_DEVICE : MOBI_E_CLAMP; // bit 1: an intra dependency never arrived
  return MOBI_OK;
}

int mobi_batch_get_planes(mobi_batch *b, int clip, int ring_idx, uint8_t *y_out, uint8_t *uv_out) {
  if (!b || clip < 0 || clip >= b->n || ring_idx < 0 || ring_idx > 5) return MOBI_E_ARG;
  if (ring_idx >= b->frames_started) return MOBI_E_NULLREF;
  HIP_TRY(hipSetDevice(b->device));
  const uint8_t *slot = b->arena + kGuard + (size_t)clip * b->clip_bytes + (size_t)((b->ring_base + 6 - ring_idx) % 6) * b->slot_bytes;
  const size_t ysz = (size_t)b->g.stride * b->g.height;
  // the planes live in HBM as macroblock tiles (mobi_tile.h); callers get the reference's row-major arrays (MD.cs:107-108, 414-415)
  if (!b->d_lin) HIP_TRY(hipMalloc((void **)&b->d_lin, b->slot_bytes));
  if (mobi_launch_untile(slot, b->d_lin, b->g.stride, b->g.height, b->stream) != 0) return MOBI_E_DEVICE;
  HIP_TRY(hipStreamSynchronize(b->stream));
  if (y_out) HIP_TRY(hipMemcpy(y_out, b->d_lin, ysz, hipMemcpyDeviceToHost));
  if (uv_out) HIP_TRY(hipMemcpy(uv_out, b->d_lin + ysz, ysz / 2, hipMemcpyDeviceToHost));
  return MOBI_OK;
}
// ---- the Bitmap of DecodeFrame(), MD.cs:260-323 ---------------------------------------------------------
static int ensure_argb(mobi_batch *b, int n_clips) {
  const size_t need = (size_t)n_clips * b->g.width * b->g.height * 4;
  if (b->d_argb && b->argb_bytes >= need) return MOBI_OK;
  if (b->d_argb) (void)hipFree(b->d_argb);
  b->d_argb = nullptr;
  b->argb_bytes = 0;
  HIP_TRY(hipMalloc((void **)&b->d_argb, need));
  b->argb_bytes = need;
  return MOBI_OK;
}
int mobi_batch_convert_argb(mobi_batch *b) {
  if (!b) return MOBI_E_ARG;
  if (b->frames_started < 1) return MOBI_E_NULLREF;
  HIP_TRY(hipSetDevice(b->device));
  if (int e = ensure_argb(b, b->n)) return e;
  MobiReconArgs a = b->args(nullptr, nullptr);
  if (mobi_launch_argb(&a, b->version, 0, b->n, b->d_argb, b->stream) != 0) return MOBI_E_DEVICE;
  b->argb_all_valid = true;
  return MOBI_OK;
}
int mobi_batch_get_argb(mobi_batch *b, int clip, uint32_t *out) {
  if (!b || clip < 0 || clip >= b->n || !out) return MOBI_E_ARG;
  if (b->frames_started < 1) return MOBI_E_NULLREF;
  HIP_TRY(hipSetDevice(b->device));
  const size_t words = (size_t)b->g.width * b->g.height;
  size_t src_clip = (size_t)clip;
  if (!b->argb_all_valid) { // convert just this clip into the front of the buffer
    if (int e = ensure_argb(b, 1)) return e;
    MobiReconArgs a = b->args(nullptr, nullptr);
    if (mobi_launch_argb(&a, b->version, clip, 1, b->d_argb, b->stream) != 0) return MOBI_E_DEVICE;
    src_clip = 0;
  }
  HIP_TRY(hipStreamSynchronize(b->stream));
  HIP_TRY(hipMemcpy(out, b->d_argb + src_clip * words, words * 4, hipMemcpyDeviceToHost));
  return MOBI_OK;
}
// the Bitmap of the frame at ring index ring_idx (0 = the newest): what DecodeFrame() returned `ring_idx` calls ago -- for callers that decode
// in groups (mobi_batch_decode_gop: frame k of a group of K sits at ring index K - 1 - k) and want every frame's Bitmap, as a converter does
// (MobiConverter/Program.cs:57-71).  (The conversion depends on Version alone, MD.cs:260-323: any frame still in the ring converts the same way.)
int mobi_batch_get_argb_at(mobi_batch *b, int clip, int ring_idx, uint32_t *out) {
  if (!b || clip < 0 || clip >= b->n || !out || ring_idx < 0 || ring_idx > 5) return MOBI_E_ARG;
  if (ring_idx >= b->frames_started) return MOBI_E_NULLREF;
  if (ring_idx == 0) return mobi_batch_get_argb(b, clip, out);
  HIP_TRY(hipSetDevice(b->device));
  const size_t words = (size_t)b->g.width * b->g.height;
  if (int e = ensure_argb(b, 1)) return e;
  b->argb_all_valid = false; // (the front of the buffer is this frame's now)
  MobiReconArgs a = b->args(nullptr, nullptr);
  a.ring_base = (b->ring_base + 6 - ring_idx) % 6;
  if (mobi_launch_argb(&a, b->version, clip, 1, b->d_argb, b->stream) != 0) return MOBI_E_DEVICE;
  HIP_TRY(hipStreamSynchronize(b->stream));
  HIP_TRY(hipMemcpy(out, b->d_argb, words * 4, hipMemcpyDeviceToHost));
  return MOBI_OK;
}
// ---- encoder-side analysis: Analyzer.InterPredict2x2 over the ring this batch keeps in HBM (Analyzer.cs:608-693) ----
int mobi_batch_motion_search(mobi_batch *b, const uint8_t *const *src_y, uint32_t *out) {
  if (!b || !src_y || !out) return MOBI_E_ARG;
  HIP_TRY(hipSetDevice(b->device));
  const size_t n = (size_t)b->n, px = (size_t)b->g.width * b->g.height, n_mbs = (size_t)b->g.mbw * b->g.mbh;
  if (int e = b->h_stage.reserve(n * px)) return e;
  if (int e = b->d_src.reserve(n * px)) return e;
  if (int e = b->d_search.reserve(n * n_mbs * 64 * 4)) return e;
  for (size_t i = 0; i < n; i++)
    if (!src_y[i]) return MOBI_E_ARG;
  b->pool->run((int)n, [&](int i) { memcpy(b->h_stage.p + (size_t)i * px, src_y[i], px); });
  HIP_TRY(hipMemcpyAsync(b->d_src.p, b->h_stage.p, n * px, hipMemcpyHostToDevice, b->stream));
  MobiReconArgs a = b->args(nullptr, nullptr);
  const int n_past = std::min(5, b->frames_started); // PastFramesY[i] == null ends the loop (:618)
  if (mobi_launch_motion_search(&a, b->d_src.p, (uint32_t *)b->d_search.p, n_past, b->stream) != 0) return MOBI_E_DEVICE;
  HIP_TRY(hipMemcpyAsync(out, b->d_search.p, n * n_mbs * 64 * 4, hipMemcpyDeviceToHost, b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));
  return MOBI_OK;
}

// ---- encoder-side forward transforms: MobiEncoder.DCT64 / DCT16 over caller buffers ----
int mobi_forward_dct(int device, int n, const int32_t *in, int32_t *out, size_t n_blocks) {
  if ((n != 8 && n != 4) || (n_blocks && (!in || !out)) || n_blocks > 0xFFFFFFFFu / 64) return MOBI_E_ARG;
  if (n_blocks == 0) return MOBI_OK;
  HIP_TRY(hipSetDevice(device));
  const size_t bytes = n_blocks * (size_t)(n * n) * 4;
  int32_t *d_in = nullptr, *d_out = nullptr;
  HIP_TRY(hipMalloc((void **)&d_in, bytes));
  if (hipMalloc((void **)&d_out, bytes) != hipSuccess) { (void)hipFree(d_in); return MOBI_E_DEVICE; }
  int rc = MOBI_OK;
  if (hipMemcpy(d_in, in, bytes, hipMemcpyHostToDevice) != hipSuccess || mobi_launch_fwd_dct(n, d_in, d_out, (uint32_t)n_blocks, nullptr) != 0 ||
      hipMemcpy(out, d_out, bytes, hipMemcpyDeviceToHost) != hipSuccess)
    rc = MOBI_E_DEVICE;
  (void)hipFree(d_in);
  (void)hipFree(d_out);
  return rc;
}

// Batches made of copies (clip c decodes the same stream as clip c mod modulus: bench.py, soak runs): how many clips' newest frame differs
// from their source clip's, compared on the device byte for byte; n_diff_out[c] (optional, n_clips entries) = differing 16-byte words of clip c.
int mobi_batch_compare_clips(mobi_batch *b, int modulus, uint32_t *n_diff_out) {
  if (!b || modulus < 1 || modulus > b->n) return MOBI_E_ARG;
  if (b->frames_started < 1) return MOBI_E_NULLREF;
  HIP_TRY(hipSetDevice(b->device));
  if (int e = b->d_search.reserve((size_t)b->n * 4)) return e;
  HIP_TRY(hipMemsetAsync(b->d_search.p, 0, (size_t)b->n * 4, b->stream));
  MobiReconArgs a = b->args(nullptr, nullptr);
  if (mobi_launch_compare_clips(&a, modulus, (uint32_t *)b->d_search.p, b->stream) != 0) return MOBI_E_DEVICE;
  std::vector<uint32_t> h(b->n);
  HIP_TRY(hipMemcpyAsync(h.data(), b->d_search.p, (size_t)b->n * 4, hipMemcpyDeviceToHost, b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));
  int bad = 0;
  for (int i = 0; i < b->n; i++) bad += h[i] != 0;
  if (n_diff_out) memcpy(n_diff_out, h.data(), (size_t)b->n * 4);
  return bad;
}

uint32_t mobi_batch_quantizer(const mobi_batch *b, int clip) {
  if (!b || clip < 0 || clip >= b->n) return 0;
  if (b->parse_mode) return b->dev_quant.empty() ? 0 : b->dev_quant[clip];
  return b->parsers[clip]->quantizer();
}
uint32_t mobi_batch_yuv_format(const mobi_batch *b, int clip) {
  if (!b || clip < 0 || clip >= b->n) return 0;
  if (b->parse_mode) return b->dev_yuvfmt.empty() ? 0 : b->dev_yuvfmt[clip];
  return b->parsers[clip]->yuv_format();
}
float mobi_batch_last_decode_ms(const mobi_batch *b) { return b ? b->last_decode_ms : 0.f; }
int mobi_batch_stride(const mobi_batch *b) { return b ? b->g.stride : 0; }
int mobi_batch_n_clips(const mobi_batch *b) { return b ? b->n : 0; }

// ---- pre-parsed replay -----------------------------------------------------------------------------
int mobi_batch_preload(mobi_batch *b, int clip, const uint8_t *data, size_t len, const uint32_t *frame_off, int n_frames, int *rc_per_frame) {
  if (!b || clip < 0 || clip >= b->n || !data || !frame_off || n_frames < 1) return MOBI_E_ARG;
  for (int f = 0; f < n_frames; f++) // every frame boundary is checked before anything of the clip's staged state is replaced
    if (frame_off[f + 1] > len || frame_off[f] > frame_off[f + 1]) return MOBI_E_ARG;
  if (b->staged.empty()) { b->staged.resize(b->n); b->staged_rc.resize(b->n); }
  b->committed = false;
  b->staged[clip] = std::make_shared<std::vector<ParsedFrame>>(n_frames);
  b->staged_rc[clip] = std::make_shared<std::vector<int>>(n_frames, MOBI_E_ARG); // "not parsed": commit never executes such a frame
  auto &dst = *b->staged[clip];
  auto &rcs = *b->staged_rc[clip];
  MobiStreamParser parser((uint32_t)b->g.width, (uint32_t)b->g.height, b->version); // fresh decoder state for this clip
  int worst = MOBI_OK;
  for (int f = 0; f < n_frames; f++) {
    int32_t off = (int32_t)frame_off[f];
    int rc = parser.parse_frame(data, frame_off[f + 1], &off, dst[f]);
    rcs[f] = rc;
    if (rc_per_frame) rc_per_frame[f] = rc;
    if (rc != MOBI_OK && worst == MOBI_OK) worst = rc;
  }
  return worst;
}
int mobi_batch_preload_clone(mobi_batch *b, int clip, int src_clip) {
  if (!b || b->staged.empty() || clip < 0 || clip >= b->n || src_clip < 0 || src_clip >= b->n || !b->staged[src_clip]) return MOBI_E_ARG;
  b->committed = false;
  b->staged[clip] = b->staged[src_clip];
  b->staged_rc[clip] = b->staged_rc[src_clip];
  return MOBI_OK;
}
int mobi_batch_commit(mobi_batch *b) {
  if (!b || b->staged.empty()) return MOBI_E_ARG;
  HIP_TRY(hipSetDevice(b->device));
  const int n = b->n;
  int nf = -1;
  for (int c = 0; c < n; c++) {
    if (!b->staged[c] || b->staged[c]->empty()) return MOBI_E_ARG; // every clip must be loaded or cloned
    if (nf < 0) nf = (int)b->staged[c]->size();
    if ((int)b->staged[c]->size() != nf) return MOBI_E_ARG;
  }
  const int n_mbs = b->g.mbw * b->g.mbh;
  const size_t desc_bytes = align_up((size_t)n * n_mbs * sizeof(MbDesc) + 8 * sizeof(MbDesc), kAlign); // slack: a wave reads up to 8 descriptors at once
  size_t cmd_bytes = 0, n_items = 0;
  b->r_plan.assign(nf, LevelPlan());
  b->r_items_off.assign(nf, 0);
  b->r_desc_off.assign(nf, 0);
  b->r_payload_off.assign(nf, 0);
  std::vector<std::vector<const ParsedFrame *>> per_frame(nf, std::vector<const ParsedFrame *>(n, nullptr));
  for (int f = 0; f < nf; f++) {
    auto &ok = per_frame[f];
    for (int c = 0; c < n; c++)
      if ((*b->staged_rc[c])[f] == MOBI_OK && (int)(*b->staged[c])[f].desc.size() == n_mbs) ok[c] = &(*b->staged[c])[f]; // never a frame no parse filled
    if (step_payload_words(ok) + kPaySlack / 4 >= ((uint64_t)1 << 32)) return MOBI_E_ARG; // 32-bit word offsets (split the batch)
    b->r_desc_off[f] = cmd_bytes;
    b->r_payload_off[f] = cmd_bytes + desc_bytes;
    cmd_bytes += desc_bytes + align_up(step_payload_words(ok) * 4 + kPaySlack, kAlign);
    b->r_plan[f].build(ok, b->g.mbw);
    b->r_items_off[f] = n_items;
    n_items += b->r_plan[f].items.size();
  }
  if (int e = b->r_cmd.reserve(cmd_bytes)) return e;
  if (int e = b->r_items.reserve(n_items * 4 + 16)) return e;
  // upload through a bounded pinned window
  const size_t win = (size_t)64 << 20;
  if (int e = b->h_stage.reserve(win)) return e;
  auto upload = [&](uint8_t *dst, const uint8_t *src, size_t bytes) -> int {
    for (size_t done = 0; done < bytes; done += win) {
      size_t chunk = std::min(win, bytes - done);
      memcpy(b->h_stage.p, src + done, chunk);
      HIP_TRY(hipMemcpyAsync(dst + done, b->h_stage.p, chunk, hipMemcpyHostToDevice, b->stream));
      HIP_TRY(hipStreamSynchronize(b->stream));
    }
    return MOBI_OK;
  };
  std::vector<uint8_t> tmp;
  for (int f = 0; f < nf; f++) {
    const size_t bytes = (f + 1 < nf ? b->r_desc_off[f + 1] : cmd_bytes) - b->r_desc_off[f];
    tmp.assign(bytes, 0);
    step_write(per_frame[f], n_mbs, (MbDesc *)tmp.data(), (uint32_t *)(tmp.data() + desc_bytes));
    if (int e = upload(b->r_cmd.p + b->r_desc_off[f], tmp.data(), bytes)) return e;
    if (!b->r_plan[f].items.empty())
      if (int e = upload(b->r_items.p + b->r_items_off[f] * 4, (const uint8_t *)b->r_plan[f].items.data(), b->r_plan[f].items.size() * 4)) return e;
  }
  b->n_frames_loaded = nf;
  b->committed = true;
  return MOBI_OK;
}
int mobi_batch_replay(mobi_batch *b, int frame_idx) {
  if (!b || !b->committed || frame_idx < 0 || frame_idx >= b->n_frames_loaded) return MOBI_E_ARG;
  HIP_TRY(hipSetDevice(b->device));
  b->ring_base = (b->ring_base + 1) % 6;
  b->step_tag = b->step_tag + 1 ? b->step_tag + 1 : 1;
  b->argb_all_valid = false;
  b->frames_started++;
  MobiReconArgs a = b->args(b->r_cmd.p + b->r_desc_off[frame_idx], b->r_cmd.p + b->r_payload_off[frame_idx]);
  return b->launch_plan(a, b->r_plan[frame_idx], (const uint32_t *)b->r_items.p + b->r_items_off[frame_idx]);
}
int mobi_batch_sync(mobi_batch *b) {
  if (!b) return MOBI_E_ARG;
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipMemcpyAsync(b->h_fault.data(), b->d_fault, sizeof(int) * b->n, hipMemcpyDeviceToHost, b->stream));
  HIP_TRY(hipMemsetAsync(b->d_fault, 0, sizeof(int) * b->n, b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));
  b->drain_events();
  for (int i = 0; i < b->n; i++)
    if (b->h_fault[i]) return (b->h_fault[i] & 2) ? MOBI_E_DEVICE : MOBI_E_CLAMP;
  return MOBI_OK;
}
uint64_t mobi_batch_cmd_bytes(const mobi_batch *b, int frame_idx) {
  if (!b || !b->committed || frame_idx < 0 || frame_idx >= b->n_frames_loaded) return 0;
  return b->r_plan[frame_idx].cmd_bytes;
}
int mobi_batch_intra_stats(const mobi_batch *b, int frame_idx, uint64_t *n_intra_mbs, uint64_t *intra_cmd_bytes) {
  if (!b || !b->committed || frame_idx < 0 || frame_idx >= b->n_frames_loaded) return MOBI_E_ARG;
  if (n_intra_mbs) *n_intra_mbs = b->r_plan[frame_idx].n_intra;
  if (intra_cmd_bytes) *intra_cmd_bytes = b->r_plan[frame_idx].intra_cmd_bytes;
  return MOBI_OK;
}
int mobi_batch_time_begin(mobi_batch *b) {
  if (!b) return MOBI_E_ARG;
  HIP_TRY(hipSetDevice(b->device));
  b->acc_ms[0] = b->acc_ms[1] = 0;
  b->acc_launches[0] = b->acc_launches[1] = 0;
  HIP_TRY(hipEventRecord(b->ev_begin, b->stream));
  return MOBI_OK;
}
int mobi_batch_time_end(mobi_batch *b, float *ms_out) {
  if (!b || !ms_out) return MOBI_E_ARG;
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipEventRecord(b->ev_end, b->stream));
  HIP_TRY(hipEventSynchronize(b->ev_end));
  HIP_TRY(hipEventElapsedTime(ms_out, b->ev_begin, b->ev_end));
  b->drain_events();
  return MOBI_OK;
}
int mobi_batch_set_kernel_timing(mobi_batch *b, int enable) {
  if (!b) return MOBI_E_ARG;
  b->ktiming = enable < 0 ? 0 : (enable > 2 ? 2 : enable);
  return MOBI_OK;
}
int mobi_batch_kernel_ms(mobi_batch *b, float *inter_ms, float *intra_ms, int *inter_launches, int *intra_launches) {
  if (!b) return MOBI_E_ARG;
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipStreamSynchronize(b->stream));
  b->drain_events();
  if (inter_ms) *inter_ms = b->acc_ms[0];
  if (intra_ms) *intra_ms = b->acc_ms[1];
  if (inter_launches) *inter_launches = b->acc_launches[0];
  if (intra_launches) *intra_launches = b->acc_launches[1];
  return MOBI_OK;
}

// ---- single stream = batch of one --------------------------------------------------------------------
mobi_dec *mobi_create(uint32_t width, uint32_t height, int version, int device) {
  mobi_batch *b = mobi_batch_create(1, width, height, version, device);
  if (!b) return nullptr;
  // the Bitmap buffer of a single stream (1.2 MB at 640x480) exists from the start: the first mobi_get_argb is then a launch and a copy like
  // every later one, not an allocation (r05: a first call that allocated right after another batch had freed 200 GB was once timed at 5 s)
  if (ensure_argb(b, 1) != MOBI_OK) { mobi_batch_destroy(b); return nullptr; }
  return new mobi_dec{b};
}
void mobi_destroy(mobi_dec *d) {
  if (!d) return;
  mobi_batch_destroy(d->b);
  delete d;
}
int mobi_decode(mobi_dec *d, const uint8_t *data, size_t len, int32_t *offset_inout) {
  if (!d || !data || !offset_inout) return MOBI_E_ARG;
  int rc = MOBI_OK;
  const uint8_t *dp[1] = {data};
  size_t lp[1] = {len};
  int e = mobi_batch_decode(d->b, dp, lp, offset_inout, &rc);
  return e != MOBI_OK ? e : rc;
}
long long mobi_selftest_div239(int device) {
  if (hipSetDevice(device) != hipSuccess) return -1;
  return mobi_launch_div239_check(nullptr);
}
int mobi_get_argb(mobi_dec *d, uint32_t *out) { return d ? mobi_batch_get_argb(d->b, 0, out) : MOBI_E_ARG; }
int mobi_get_planes(mobi_dec *d, int ring_idx, uint8_t *y_out, uint8_t *uv_out) { return d ? mobi_batch_get_planes(d->b, 0, ring_idx, y_out, uv_out) : MOBI_E_ARG; }
int mobi_stride(const mobi_dec *d) { return d ? d->b->g.stride : 0; }
uint32_t mobi_quantizer(const mobi_dec *d) { return d ? mobi_batch_quantizer(d->b, 0) : 0; }
uint32_t mobi_yuv_format(const mobi_dec *d) { return d ? mobi_batch_yuv_format(d->b, 0) : 0; }
uint32_t mobi_width(const mobi_dec *d) { return d ? (uint32_t)d->b->g.width : 0; }
uint32_t mobi_height(const mobi_dec *d) { return d ? (uint32_t)d->b->g.height : 0; }

} // extern "C"
