// mobi_parse.h -- host half of one decoder instance: serial bitstream parse -> command list.
//
// Mirrors the parse side of LibMobiclip.Codec.Mobiclip.MobiclipDecoder (MobiclipDecoder.cs,
// "MD.cs"): bit reader (:2970-3015), frame headers (:113-143, :224-236), MV prediction
// (:163-208), the partition tree (:469-1746), residual CBP/VLC (:1818-1833, :2909-2968,
// :3330-3432), intra macroblock syntax (:1759-1880, :2776-2902) and the quantiser tables
// (:3884-3925).  It writes no pixels: every reconstruction step becomes a command (mobi_cmd.h).
#ifndef MOBI_PARSE_H
#define MOBI_PARSE_H
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <vector>

#include "mobi_cmd.h"
#include "mobi_state.h"

enum { MOBI_INTRA_CLASSES = 16 }; // launch classes of intra macroblocks inside a dependency level (finish_levels)
struct ParsedFrame {
  FrameHdr hdr;
  std::vector<MbDesc> desc;
  std::vector<uint32_t> payload;
  std::vector<uint32_t> intra_mbs;   // MB indices, grouped by level (ascending), by class inside a level (class_start), raster order inside a class
  std::vector<uint32_t> level_start; // size n_levels+2: intra_mbs[level_start[L] .. level_start[L+1]) for L = 1..n_levels
  // The same macroblocks as launch items (mobi_recon_intra; LevelPlan in mobi_abi.cpp), four words each, in the order of intra_mbs:
  //   [0] mb   [1] MbDesc.w1   [2] MbDesc.payload_off (inside this clip's payload)   [3] flags: [0] 16x16 plane present, [1] has intra
  //   dependencies, [2] has intra dependents, [3] at the picture's left / right / top edge, [14:5] level words, [31:16] plane parameter.
  // Written by the parser while the descriptors are in its cache: the step's launch list is then a concatenation, not 60 random
  // reads per clip into descriptor tables other cores wrote (8 ms per step of 1024 clips, measured r04).
  std::vector<uint32_t> intra_items;
  std::vector<uint32_t> class_start; // [L * MOBI_INTRA_CLASSES + class] -> first index of that class inside level L (class: mobi_parse.cpp, finish_levels); one more entry = the end
  void clear();
  size_t cmd_bytes() const { return desc.size() * sizeof(MbDesc) + payload.size() * 4; }
};

// Geometry helpers shared by the parser (dependency levels) and the kernels (availability):
// which macroblock owns the pixel at linear address `a` of the Y / UV plane; -1 = padding / outside.
struct MobiGeom {
  int width, height, stride, mbw, mbh;
  int lg; // log2(stride): the stride is 256 / 512 / 1024 (MD.cs:50-52), so rows and columns come by shift and mask
  int owner_luma(long a) const {
    if (a < 0) return -1;
    const long row = a >> lg, col = a & (stride - 1);
    if (col >= width || row >= height) return -1;
    return (int)((row >> 4) * mbw + (col >> 4));
  }
  int owner_chroma(long a) const {
    if (a < 0) return -1;
    const long row = a >> lg, col = a & (stride - 1);
    const long x = col >= stride / 2 ? col - stride / 2 : col;
    if (x >= width / 2 || row >= height / 2) return -1;
    return (int)((row >> 3) * mbw + (x >> 3));
  }
};
// halo the intra kernel loads around a macroblock (must match mobi_kernels.hip)
enum { MOBI_HALO_Y_RIGHT = 23, MOBI_HALO_C_RIGHT = 15 };

void mobi_build_scale_table(int q, int32_t out[MOBI_SCALE_STRIDE]);
// directional intra predictors as four tile offsets per sample (see mobi_parse.cpp); out: MOBI_TAP_ENTRIES x 4 int16; false = self-check failed
bool mobi_build_intra_taps(int16_t *out, int pitch);

// Streams the reference decodes and this library refuses (MOBI_E_UNSUPPORTED), by cause; process-wide counters, a measuring aid.  r05: one
// cause is left -- the others are decoded (mc_leaf: vectors are normalised; resid_block: the transforms' scratch is kept; intra_*: wide
// plane parameters travel behind the level words).
enum { MOBI_REFUSE_MV = 0,     // (r01-r04: |MV| > MOBI_MV_LIMIT half-pels; r05: nothing counts here any more)
       MOBI_REFUSE_QUANT = 1,  // (r03: ModsDS quantiser < 12; r04 decodes those frames: nothing counts here any more)
       MOBI_REFUSE_RUN = 2,    // a walk through Internal[] (MD.cs:3424-3429) left a coefficient outside int16 in a block whose residual stays within
                               // +-319 everywhere (so that the clamp table need not fault, MobiConst.cs:587): a literal travels in the level's 16 bits
       MOBI_REFUSE_PLANE = 3,  // (r01-r04: a plane-predictor parameter outside int16; r05: nothing counts here any more)
       MOBI_REFUSE_CLASSES = 4 };
extern std::atomic<unsigned long> mobi_refusal_count[MOBI_REFUSE_CLASSES];
extern std::atomic<unsigned long> mobi_scratch_read_count;
extern std::atomic<unsigned long> mobi_literal_frame_count; // frames shipped as literal values (MobiStreamParser::literal_frame): a measuring aid too

class MobiStreamParser {
 public:
  MobiStreamParser(uint32_t width, uint32_t height, int version);
  // d.Data=data; d.Offset=*offset; DecodeFrame(); *offset=d.Offset.  Returns MOBI_OK or MOBI_E_*.
  // On error `out` is not to be executed; the ring still advances (MD.cs:102-108 ran already).
  int parse_frame(const uint8_t *data, size_t len, int32_t *offset, ParsedFrame &out);

  // The decoder state that survives a frame, as the device parsers keep it (mobi_state.h): a clip whose frame the device parser could not
  // finish is parsed again here from the state it had when that frame started, and stays with this parser from then on (mobi_abi.cpp).
  void import_state(const MobiDevState &st, const MobiDevTail &tail);
  void export_state(MobiDevState &st, MobiDevTail &tail);
  uint32_t internal_word(uint32_t idx); // Internal[idx] as the reference holds it between frames, 10 <= idx < 392 (tests: against the oracle's)
  // The last frame was one the device parsers would have finished too (no walk, no token without a level, no value beyond their fields),
  // and nothing a walk once wrote behind the MV row cache is left: the clip may go back to the device parsers (mobi_abi.cpp, dp_return).
  bool device_ready() const;

  uint32_t quantizer() const { return quant_; }
  uint32_t yuv_format() const { return yuvfmt_; }
  int frames_started() const { return frames_started_; }
  const uint8_t *mode_cache() const { return mcache_; }                // bytes of Internal[0..9] (40 of them)
  bool quant_tables_set() const { return (dq8_[1] & 0xFF) != 0; }      // SetupQuantTables ran: the zigzag bytes of Internal[10..] are there
  const MobiGeom &geom() const { return g_; }
  int version() const { return version_; }

 private:
  struct Err { int code; };
  [[noreturn]] void fail(int code) const { throw Err{code}; }
  // MOBI_E_UNSUPPORTED by cause (INTEGRATION.md error table): counted for tools/exp_refusals.py
  [[noreturn]] void refuse(int cause) const { mobi_refusal_count[cause].fetch_add(1, std::memory_order_relaxed); throw Err{-6 /* MOBI_E_UNSUPPORTED */}; }
  // bit reader
  uint32_t data_u16(long off) const;
  void fill_bits();
  void take(int n);
  uint32_t ue();
  int se();
  // syntax
  void setup_quant(uint32_t q);
  void parse_p(ParsedFrame &out);
  void parse_i(ParsedFrame &out);
  void pblock(int wi, int hi, int x, int y, int mv_slot);
  void mc_leaf(int wi, int hi, int x, int y, int ref, int dx, int dy, int mv_slot);
  void build_cells();
  void check_window(long pos, int w, int h, int phase, long plane_len) const;
  void p_residual();
  void resid_area(int area);
  void resid_block(int area, int sub, bool is8);
  uint32_t internal_read(uint32_t idx);
  void internal_write(uint32_t idx, uint32_t v);
  void scratch_materialise();
  void build_dq();
  void literal_frame(ParsedFrame &out);
  bool surely_faults(bool is8, int variant) const;
  void intra_full();
  void intra_sub();
  void intra_chroma(uint32_t cbp);
  void intra_area_fixed(int area, int mode, bool coded);
  int pmode(int ci, bool four);
  uint32_t plane_param(int p, int r);
  void check_intra_reads(int mode, long off, bool four) const;
  long area_offset(int area, int sub) const;
  void begin_mb(int mb, int type);
  void end_mb();
  void finish_levels(ParsedFrame &out);

  MobiGeom g_;
  int version_, ver_; // ver_: table index 0 = Moflex3DS, 1 = ModsDS
  // stream
  const uint8_t *data_ = nullptr;
  long len_ = 0;
  int off_ = 0;
  uint32_t win_ = 0; // r3
  int nbr_ = 0;      // nrBitsRemaining
  // persistent decoder state
  uint32_t quant_ = 0, yuvfmt_ = 0;
  uint32_t tq_ = MOBI_TQ_NONE; // the quantiser dq8_ / dq4_ were built for: SetupQuantizationTables assigns Quantizer before its table index can
                               // throw (MD.cs:3886-3890), so after a throw the OLD tables serve the new Quantizer (ModsDS, q >= 54)
  uint32_t dq8_[64], dq4_[16]; // Internal[10..73], Internal[74..89]
  uint8_t mcache_[40];         // bytes of Internal[0..9]
  int vlc_table_ = 0;          // Internal[218] == 1
  // r04: the part of Internal[] that a coefficient run past its block, or a ModsDS quantiser below 12, reads and writes (MD.cs:3424-3429):
  // the coefficient block Internal[90..153] as every residual block and every transform variant leaves it, the table select as a word,
  // and whatever was written behind the MV row cache.  r05: the transforms' scratch Internal[154..217] too, lazily -- the coefficients of the
  // last full 8x8 transform and of the last 16-coefficient one behind it are kept, and their first passes are made when a walk reads or
  // writes there (scratch_materialise).
  uint32_t ib_[64] = {0};      // Internal[90..153]
  uint32_t scr_[64] = {0};     // Internal[154..217] as of the last scratch_materialise()
  uint32_t sc64_[64], sc16_[16]; // coefficients of the pending transforms (pend64_: IDCT64Px8 -> all of scr_; pend16_: IDCT16Px8 -> scr_[0..31], behind it)
  bool pend64_ = false, pend16_ = false;
  uint32_t i218_ = 0;          // Internal[218]
  uint32_t itail_[392] = {0};  // Internal[idx] for indices that are nothing else (behind the MV row cache)
  bool frame_literal_ = false; // a block of this frame read or wrote Internal[] out of its place: its residuals ship as literal values
  bool frame_fault_ = false;   // ... and one of them holds a coefficient beyond int16 whose transform must leave the clamp table's domain
  bool frame_host_only_ = false; // this frame holds something the device parsers stop at although it decodes (dp_return: the clip stays here)
  bool last_frame_ok_ = false;
  bool big_unsure_ = false;    // an ordinary block of this frame holds such a coefficient and need NOT leave it (surely_faults)
  int frames_started_ = 0;
  std::vector<int> mvc_; // MV row cache, Internal[221..]
  int predx_ = 0, predy_ = 0;
  // current frame / MB being built
  ParsedFrame *out_ = nullptr;
  int cur_mb_ = 0, cur_x_ = 0, cur_y_ = 0;
  long cur_off_ = 0;
  uint32_t leaves_[128], coefs_[384]; // MC leaves (two words each) and residual levels (6 areas x 64) of the macroblock being built
  int n_leaf_words_ = 0, n_coefs_ = 0;
  uint32_t recs_[MOBI_INTRA_RECORDS];
  uint32_t cells_[MOBI_MV_CELLS];
  uint32_t cbp6_ = 0, t8mask_ = 0, w3_ = 0;
  int32_t wide_[MOBI_WIDE_PARAMS];      // plane parameters outside int16 of the macroblock being built (mobi_cmd.h)
  bool any_wide_ = false;
  int mb_type_ = 0;
};

#endif
