/*
 * mobiclip_hip.h -- C ABI of libmobiclip_hip.so: MI355X (gfx950) Mobiclip frame reconstruction.
 *
 * The reference has no FFI seam; the seam is the public surface of
 *   LibMobiclip.Codec.Mobiclip.MobiclipDecoder   (MobiclipDecoder.cs:13-61, "MD.cs")
 * whose callers do   d.Data = frame; d.Offset = o; d.DecodeFrame(); ... d.Offset ...
 * (MobiConverter/Program.cs:69-71,243-250,393-395; MobiclipDecoder/Form1.cs:261-263,292-302,
 * 359-364,463-465,525-527).  Each entry point below names the member it replaces; the C#
 * binding a maintainer would add is in INTEGRATION.md.
 *
 * Split of work: the serial VLC / Exp-Golomb parse (MD.cs bit reader and syntax, :113-259,
 * :469-3432) runs on the host inside these calls and emits a flat per-macroblock command list;
 * dequant + inverse transforms, intra prediction, motion compensation and residual add
 * (MD.cs:418-456, :1883-2774, :3017-3327, :3435-3798) run as HIP kernels (one wavefront per eight
 * adjacent inter macroblocks, one per four intra macroblocks).  There is NO CPU reconstruction path in this library: without a HIP device every
 * create call fails.
 */
#ifndef MOBICLIP_HIP_H
#define MOBICLIP_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default) /* the library is built with -fvisibility=hidden: these declarations are all it exports */
#endif

/* MobiclipDecoder.MobiclipVersion, MD.cs:32-37 */
#define MOBI_VERSION_VXDS 0      /* unimplemented stub in the reference too (MD.cs:63-95) */
#define MOBI_VERSION_MODSDS 1
#define MOBI_VERSION_MOFLEX3DS 2

/* Return codes.  0 = frame decoded.  Negative = the reference would have thrown inside
 * DecodeVXS2 and returned a null Bitmap (MD.cs:325-328); the code says which check fired. */
#define MOBI_OK 0
#define MOBI_E_INDEX (-1)        /* managed array bounds: MC source window, intra neighbour at a negative
                                    offset, VLC/CBP/quantizer table index, bitstream read past Data */
#define MOBI_E_NULLREF (-2)      /* reference frame slot Y[ref] never decoded (MD.cs:413) */
#define MOBI_E_PARTCODE (-3)     /* illegal partition code: explicit throw (e.g. MD.cs:625,730,827) */
#define MOBI_E_VERSION (-4)      /* VxDS / unknown version */
#define MOBI_E_CLAMP (-5)        /* residual add left the clamp table domain (MobiConst.cs:587; MD.cs:3551);
                                    detected on the GPU, so the parser state has already advanced */
#define MOBI_E_UNSUPPORTED (-6)  /* the reference decodes this frame to something and this library refuses.  r05: ONE input is left -- a
                                    coefficient run that walks through `Internal` (MD.cs:3424-3429) and leaves a coefficient beyond
                                    int16 in a block whose residual nevertheless stays within +-319 everywhere (so that the clamp
                                    table need not fault): sums that cancel through 32-bit wrap-around; never seen in 40 000 corrupted
                                    frames.  Everything r01-r04 refused is decoded, wherever the parse runs (error table) */
#define MOBI_E_ARG (-7)          /* bad argument / dimensions not a multiple of 16 (the reference cannot
                                    decode those either: MD.cs:216-217) */
#define MOBI_E_DEVICE (-8)       /* HIP error (no device, allocation, launch) */

typedef struct mobi_dec mobi_dec;     /* one stream: replaces one MobiclipDecoder instance */
typedef struct mobi_batch mobi_batch; /* N independent streams of equal geometry, decoded in lock step */

/* ---- single stream ------------------------------------------------------------------------ */
/* new MobiclipDecoder(Width, Height, Version)  (MD.cs:41-54).  device = HIP ordinal. */
mobi_dec *mobi_create(uint32_t width, uint32_t height, int version, int device);
void mobi_destroy(mobi_dec *d);
/* d.Data = data (len bytes); d.Offset = *offset_inout; d.DecodeFrame(); *offset_inout = d.Offset
 * (MD.cs:56-61, 97-259).  Synchronous: returns after the frame is reconstructed on the device. */
int mobi_decode(mobi_dec *d, const uint8_t *data, size_t len, int32_t *offset_inout);
/* d.Y[ring_idx] / d.UV[ring_idx]  (MD.cs:19-20), copied in the reference layout:
 * y_out: Stride*Height bytes; uv_out: Stride*Height/2 bytes, U in columns [0,Stride/2), V in
 * [Stride/2,Stride) of each row (MD.cs:414-415).  Either pointer may be NULL.
 * Returns MOBI_E_NULLREF when that ring slot has never been produced.  After a call that returned an error, slot 0 of that
 * stream is not the reference's partial picture (a frame that failed in the parse is not reconstructed at all: the slot still
 * holds the picture of six frames earlier); and P-frames that predict from it differ from the reference's (which predict from its
 * partial picture) until the next I-frame, from which on both are identical again. */
int mobi_get_planes(mobi_dec *d, int ring_idx, uint8_t *y_out, uint8_t *uv_out);
/* The Bitmap that DecodeFrame() returns (MD.cs:260-323) for the frame just decoded: width*height 0xAARRGGBB
 * words, row pitch = width (Format32bppArgb as LockBits hands it out: bytes B,G,R,A).  Chroma is averaged from up
 * to four neighbours by pixel parity, then the float matrix with range stretch (Moflex3DS) or the integer form
 * (ModsDS); float arithmetic is IEEE single, one rounding per operator of the source, no FMA.
 * MOBI_E_NULLREF before the first frame. */
int mobi_get_argb(mobi_dec *d, uint32_t *out);
/* Self-test of the Bitmap kernel's arithmetic: its three-instruction x / 239f against the correctly rounded IEEE
 * division for every float bit pattern with 1e-30 <= |x| <= 1e30 (about 20 ms on the device).  Returns the number of
 * mismatches (0 is the only acceptable answer) or -1 when the device cannot be used. */
long long mobi_selftest_div239(int device);
int mobi_stride(const mobi_dec *d);            /* d.Stride     (MD.cs:30,50-52) */
uint32_t mobi_quantizer(const mobi_dec *d);    /* d.Quantizer  (MD.cs:26) */
uint32_t mobi_yuv_format(const mobi_dec *d);   /* d.YuvFormat  (MD.cs:27) */
uint32_t mobi_width(const mobi_dec *d);        /* d.Width      (MD.cs:17) */
uint32_t mobi_height(const mobi_dec *d);       /* d.Height     (MD.cs:18) */

/* ---- batch of independent clips (the throughput path) -------------------------------------- */
/* N decoder instances that share geometry/version; they share nothing else (MD.cs:15-39).
 * Tuning knobs read from the environment when a batch is created (none changes a result: rc, Offset, Quantizer and planes do not depend
 * on where a clip is parsed -- see mobi_batch_set_parse_mode): MOBI_PARSE_THREADS (host parse pool; default
 * one per two hardware threads, at most 64), MOBI_DEVICE_PARSE and MOBI_HYBRID_HOST_CLIPS (below), MOBI_FUSED_STEP_MBS (a frame step of
 * at most this many macroblocks is ONE launch, mobi_recon_step; default 256 x 1200, 0 = always two launches). */
mobi_batch *mobi_batch_create(int n_clips, uint32_t width, uint32_t height, int version, int device);
void mobi_batch_destroy(mobi_batch *b);
/* One DecodeFrame() per clip: data[i]/len[i]/offsets[i] as in mobi_decode; rc[i] per clip.
 * Parses on the host, uploads the command lists, launches, synchronises.  Returns MOBI_OK or a
 * MOBI_E_DEVICE/ARG failure of the call itself (per-clip stream errors go to rc[]). */
int mobi_batch_decode(mobi_batch *b, const uint8_t *const *data, const size_t *len, int32_t *offsets, int *rc);
/* Where mobi_batch_decode parses the bitstreams.  0: host threads, command lists uploaded per call.  1: on the GPU, one
 * wavefront per clip (mobi_dparse.hip): Data[Offset..) of every clip is uploaded instead and the command lists never leave
 * HBM; same rc / Offset / planes.  The parse of one clip is serial and a GPU lane is slow at it; the GPU wins by running
 * thousands of clips at once, from about 20 resident clips per host parse thread upward.  Default: by batch size (device parse from max(640, 20 x threads) clips,
 * unless the first call hands over far more than a frame per clip -- whole files as Data, MOC5 style -- which the device path
 * would have to upload again for every frame), or MOBI_DEVICE_PARSE=0/1/2/3.  2 = hybrid: the GPU parses most clips while the
 * host pool parses a fixed share of them (a fifth, at most 1024; MOBI_HYBRID_HOST_CLIPS) at the same time; one set of
 * reconstruction launches serves both.  3 = as 1 with the lock-step parser in front of EVERY step (mobi_lsparse.hip: a few clips per
 * wavefront, one per lane, all lanes in one instruction stream): it finishes the frames that decode without incident -- identically, word
 * for word -- and leaves every other clip to the one-wavefront-per-clip parser, which then runs for those alone.  By default the parser in
 * front is chosen step by step: the lock-step parser from 5120 clips (23 ms per 640x480 P-frame step of 24576 clips against 57) and for
 * I-frame steps from 768 clips (15 ms against 30 at 4096 clips); 1 = never.
 * THE RESULT DOES NOT DEPEND ON THE MODE (r05).  The device parsers finish the frames that decode without incident.  A frame they cannot
 * finish -- anything the reference throws on, a coefficient run that walks through `Internal` (MD.cs:3424-3429), a ModsDS quantiser below
 * 12, a value the command list has to escape -- is parsed again by the host parser inside the same call (mobi_batch_wait for asynchronous
 * steps), from the decoder state the clip had when that frame started, which the device keeps for exactly this; the clip then stays with the
 * host parser (mobi_batch_host_clips counts them), parsed beside the GPU's clips as the hybrid mode's share is, until it has had a run of
 * frames the device parsers finish (4, doubled with every hand-over, at most 256), and goes back with the host parser's state.
 * Can only be changed before the first frame: the decoder state lives on one side. */
int mobi_batch_set_parse_mode(mobi_batch *b, int device_parse);
/* how many clips of the batch the host parser parses at present: all of them in mode 0; in the other modes the hybrid share plus the clips
 * that have had a frame the device parsers could not finish and have not gone back yet */
int mobi_batch_host_clips(const mobi_batch *b);
/* Parse mode 3: how many clips of the last finished frame step the lock-step parser finished itself (the rest went to the other one);
 * -1 in the other modes or before the first step. */
int mobi_batch_lockstep_finished(const mobi_batch *b);
/* Asynchronous frame steps, for callers that already hold the next frame of every clip (demuxed Moflex / Mods packets: the Offset
 * to start from does not depend on the previous frame's parse).  The batch must parse on the GPU (the default for large batches, see above;
 * mobi_batch_set_parse_mode(b, 1), 2 or 3 otherwise: the hybrid mode's host share is parsed inside mobi_batch_submit) and at most two steps
 * may be in flight.
 *   mobi_batch_submit: copies the bytes data[i][offsets[i] .. len[i]) of every clip into pinned memory and enqueues upload, parse and
 *                      reconstruction of one frame step behind the step before; returns without waiting for the GPU.  The caller's
 *                      buffers may be reused as soon as it returns.  The upload has a stream of its own, and so has the lock-step parser
 *                      (parse mode 3): the parse of step n + 1 then runs under the reconstruction of step n (its output buffers
 *                      belong to the step).
 *   mobi_batch_wait:   waits for the OLDEST step in flight and reports what mobi_batch_decode would have: rc[i] per clip,
 *                      offsets_out[i] = Offset after the frame (may be NULL).
 * mobi_batch_decode, mobi_batch_get_planes and the other calls that read results wait for everything enqueued; mobi_batch_decode is
 * refused (MOBI_E_ARG) while steps are in flight.
 * WHICH FRAME IS WHERE: mobi_batch_submit turns the ring at once (Y[i] = Y[i-1], MD.cs:102-106, happens at submission, not at
 * completion).  With steps in flight, ring_idx 0 of mobi_batch_get_planes / get_argb / convert_argb is the NEWEST submitted step's
 * frame, and the frame of the step mobi_batch_wait has just reported sits at ring_idx = mobi_batch_in_flight(b) (1 while one later
 * step is in flight).  The getters wait for everything enqueued before they copy, so reading results between submit and wait drains the
 * pipeline: read after the last wait, or accept the drain.  mobi_batch_quantizer / mobi_batch_yuv_format describe the step last
 * WAITED for.  If mobi_batch_submit fails after it has started to enqueue (MOBI_E_DEVICE), the batch is drained and refuses all
 * further steps: destroy it.  A frame the device parsers could not finish is repaired in mobi_batch_wait (every such clip of the step
 * at once: their start states in one copy, their parses on the host threads, one reconstruction of those clips per affected step); rc,
 * Offset, Quantizer and the frame's planes are then the host parser's, as in mobi_batch_decode.  ONE thing is unspecified in this mode
 * only: when such a clip's NEXT frame was already in flight (parsed by the device from the state the failed frame left and reconstructed
 * before anybody knew), that ring slot may hold what the wrong parse wrote -- so if the repaired parse of that next frame is itself
 * rejected (rc != MOBI_OK) the slot's content is unspecified (mobi_batch_decode leaves the picture of six frames earlier there), and a
 * repaired frame that predicts from the OLDEST ring picture (reference 5, the slot the step in flight was writing) may predict from it.
 * Both need a damaged stream; intact streams, and damaged ones under mobi_batch_decode / mobi_batch_decode_gop, are exact.
 * What it buys: the host gathers and uploads step n + 1 while the GPU parses step
 * n, and the GPU goes from parse to reconstruction without asking the host for launch sizes. */
int mobi_batch_submit(mobi_batch *b, const uint8_t *const *data, const size_t *len, const int32_t *offsets);
int mobi_batch_wait(mobi_batch *b, int32_t *offsets_out, int *rc);
int mobi_batch_in_flight(const mobi_batch *b); /* steps submitted and not yet waited for: 0, 1 or 2 */
/* Frame-parallel groups (r06): K = n_frames consecutive frames of EVERY clip in one call, 1 <= K <= 6 (the ring holds six pictures,
 * MD.cs:19-20: every frame of the group is still there when the call returns -- frame k of the group at ring_idx K - 1 - k).  For callers
 * that hold whole packets of a clip's next frames -- what the Moflex / Mods demuxers deliver (MoLiveDemux.cs:349-358, ModsDemuxer.cs:97-117);
 * not for MOC5-style callers whose next Offset is only known once the frame before has been parsed.  All arrays are [k * n_clips + c]:
 * frame k of clip c is data[..][offsets[..] .. len[..]), exactly what K calls of mobi_batch_decode would have been given; rc[..] and the
 * Offsets after the frames come back the same way, and rc, Offset, Quantizer and planes ARE what those K calls give (tests: fuzzed
 * streams, every parse mode).  What it buys: a frame's PARSE needs from the frame before it only what the frame headers determine
 * (Quantizer, YuvFormat, the frame count: MD.cs:113-143, 224-236, 3884-3925; measured, profiles/r06_framedep.txt), so the device parsers
 * run over n_clips * K frames side by side -- and their cost per frame falls with the number of lanes they are given -- while the
 * reconstruction stays one step per frame, in order.  A start state predicted wrong, like a frame the device parsers do not finish,
 * hands that clip's remaining frames to the host parser inside the same call (the prediction can cost time, never the result).
 *   mobi_batch_decode_gop: the whole group, synchronously.  A batch that parses on the host (small batches; mobi_batch_set_parse_mode(b, 0))
 *                          simply runs its K steps one after the other.
 *   mobi_batch_gop_begin / mobi_batch_gop_finish: the same in two halves, for callers that keep the GPU fed: begin gathers and uploads a
 *                          group (and starts its parse when no group is in front of it) and returns; finish reports the OLDEST group
 *                          begun and not finished.  At most two groups may be begun: with group g + 1 begun before group g is finished,
 *                          its upload runs beside group g's reconstruction and its parse goes out behind the steps of g's last part:
 *                          the GPU parses while the caller gathers the group after next.  The batch must parse on the GPU (as for
 *                          mobi_batch_submit).  The ring turns in finish, once per frame; planes are read after finish.
 *                          A group begun this way may hold up to 128 frames -- what is parsed side by side is not bound by the ring, and
 *                          a small batch fills the parsers' lanes only with that many -- and finish hands them out SIX AT A TIME, oldest
 *                          first: every call reconstructs and reports min(6, mobi_batch_gop_frames_pending(b)) frames of the oldest group
 *                          (rc / offsets_out [j * n_clips + c], j counted from the part's first frame; that part's frame j sits at ring
 *                          index part_size - 1 - j afterwards), so a group of 12 is finished by two calls, one of 32 by six, one of 128 by 22.
 *                          What a group writes lives in HBM until it is reconstructed (per frame of it about 40 KB of descriptors and
 *                          items and the worst-case payload part, 0.3 - 0.5 MB at 640x480): a group that does not fit is refused by
 *                          mobi_batch_gop_begin (the allocator's error, nothing enqueued, the batch goes on).
 * mobi_batch_decode / mobi_batch_submit are refused (MOBI_E_ARG) while a group is begun and not finished. */
int mobi_batch_decode_gop(mobi_batch *b, int n_frames, const uint8_t *const *data, const size_t *len, int32_t *offsets, int *rc);
int mobi_batch_gop_begin(mobi_batch *b, int n_frames, const uint8_t *const *data, const size_t *len, const int32_t *offsets);
int mobi_batch_gop_finish(mobi_batch *b, int32_t *offsets_out, int *rc);
int mobi_batch_gop_in_flight(const mobi_batch *b); /* groups begun and not yet finished: 0, 1 or 2 */
int mobi_batch_gop_frames_pending(const mobi_batch *b); /* frames of the oldest such group that no finish has reported yet (0: none begun) */
/* Wall-clock milliseconds the last mobi_batch_decode call spent inside the library (parse or upload, launches, sync). */
float mobi_batch_last_decode_ms(const mobi_batch *b);
int mobi_batch_get_planes(mobi_batch *b, int clip, int ring_idx, uint8_t *y_out, uint8_t *uv_out);
/* Bitmaps (MD.cs:260-323): mobi_batch_convert_argb converts ring slot 0 of EVERY clip into a device-resident buffer
 * (asynchronously, on the batch's stream); mobi_batch_get_argb copies one clip's width*height words out, converting
 * just that clip first if the whole-batch conversion has not been run for the current frame. */
int mobi_batch_convert_argb(mobi_batch *b);
int mobi_batch_get_argb(mobi_batch *b, int clip, uint32_t *out);
/* ... of the frame at ring index ring_idx (0 = the newest; MOBI_E_NULLREF for a slot never produced): the Bitmap DecodeFrame() returned
 * ring_idx calls ago.  For callers that decode in groups and want every frame's Bitmap: frame k of a group of K is ring index K - 1 - k. */
int mobi_batch_get_argb_at(mobi_batch *b, int clip, int ring_idx, uint32_t *out);
/* Encoder-side analysis: Analyzer.InterPredict2x2 (Analyzer.cs:608-681) for every 2x2 luma block of
 * every macroblock of every clip, as SolveInterPredictionPuzzle calls it (:683-693): three-step search (6, 3, 1 pels) in up
 * to five past frames = ring slots 0..4 of this batch (the encoder's PastFramesY, MobiEncoder.cs:138-144).
 * src_y[clip]: the picture to analyse, width*height luma bytes, pitch = width.  out[((clip * n_mbs + mb) * 64) + Y*8 + X] =
 * (Delta.X & 0xFF) | (Delta.Y & 0xFF) << 8 | Frame << 16 | score << 20 (Delta in half pels, as the reference stores it;
 * score = 0xFFF when the ring is empty). */
int mobi_batch_motion_search(mobi_batch *b, const uint8_t *const *src_y, uint32_t *out);
/* The encoder's forward transforms: MobiEncoder.DCT64 (Encoder/MobiEncoder.cs:962-1010, n = 8) and DCT16
 * (:1146-1178, n = 4) of n_blocks residual blocks (Block - CompVals, Encoder/MacroBlock.cs:584-588), n*n int32 each, back to back;
 * out as the reference returns it (the second pass stores transposed).  Integer arithmetic, truncating divisions: bit-exact.
 * The quantiser behind it (float division + Math.Round, MacroBlock.cs:591-595) is not part of this library. */
int mobi_forward_dct(int device, int n, const int32_t *in, int32_t *out, size_t n_blocks);
/* For batches made of copies (clip c was given the same stream as clip c mod modulus: a benchmark, a soak run): compares the newest frame
 * (ring slot 0) of EVERY clip with that of its source clip on the device, byte for byte, and returns how many clips differ (0 = none; < 0 =
 * MOBI_E_*).  n_diff_out, if not NULL, receives the number of differing 16-byte words per clip (n_clips entries).  Checking the `modulus`
 * source clips against the reference decoder then vouches for all of them (the reference has no counterpart: one decoder, one stream). */
int mobi_batch_compare_clips(mobi_batch *b, int modulus, uint32_t *n_diff_out);
uint32_t mobi_batch_quantizer(const mobi_batch *b, int clip);
uint32_t mobi_batch_yuv_format(const mobi_batch *b, int clip);
int mobi_batch_stride(const mobi_batch *b);
int mobi_batch_n_clips(const mobi_batch *b);

/* Pre-parsed replay: keep the command lists of whole clips resident in HBM so that a timed region
 * contains reconstruction only (SURVEY.md 8(d) "GPU timing").
 *  preload : parse all n_frames of `clip` (frame f = data[frame_off[f] .. frame_off[f+1])) and
 *            stage their command lists on the host; rc per frame optional.
 *  preload_clone: give `clip` a private copy of another clip's staged command lists (its own HBM
 *            bytes, same content) -- lets a benchmark run more clips than it generated streams.
 *  commit  : upload everything staged; builds the per-level launch lists across clips.
 *  replay  : one frame step for every clip: ring rotate + reconstruction kernels for frame
 *            `frame_idx`, asynchronously on the batch stream.  No host parse, no H2D.
 *  sync    : wait for the stream; returns MOBI_E_CLAMP if any clip flagged a clamp-domain fault. */
int mobi_batch_preload(mobi_batch *b, int clip, const uint8_t *data, size_t len, const uint32_t *frame_off,
                       int n_frames, int *rc_per_frame);
int mobi_batch_preload_clone(mobi_batch *b, int clip, int src_clip);
int mobi_batch_commit(mobi_batch *b);
int mobi_batch_replay(mobi_batch *b, int frame_idx);
int mobi_batch_sync(mobi_batch *b);
/* command-list bytes the kernels read for frame `frame_idx`, summed over clips (roofline accounting) */
uint64_t mobi_batch_cmd_bytes(const mobi_batch *b, int frame_idx);
/* of frame `frame_idx`, summed over clips: intra macroblocks, and the command-list bytes that belong to them (descriptor, block
 * records, level words).  mobi_recon_inter8 neither reads those bytes nor touches those macroblocks' pixels: bench.py leaves them
 * out of its algorithmic bytes */
int mobi_batch_intra_stats(const mobi_batch *b, int frame_idx, uint64_t *n_intra_mbs, uint64_t *intra_cmd_bytes);
/* milliseconds between two internally recorded HIP events bracketing the replay launches of the last
 * `mobi_batch_time_begin` .. `mobi_batch_time_end` region, measured on the batch's own stream */
int mobi_batch_time_begin(mobi_batch *b);
int mobi_batch_time_end(mobi_batch *b, float *ms_out);
/* per-kernel-class device time (ms) accumulated since the last time_begin: [0] inter MC+IDCT kernel,
 * [1] intra kernel launches; measured with HIP events around launches on the batch's stream.
 * level 0: off; 1: the inter launches only (the dominant kernel -- what a roofline needs; the events themselves cost a
 * few microseconds per launch); 2: every launch */
int mobi_batch_set_kernel_timing(mobi_batch *b, int level);
int mobi_batch_kernel_ms(mobi_batch *b, float *inter_ms, float *intra_ms, int *inter_launches, int *intra_launches);

const char *mobi_error_string(int rc);
/* library self-description: "libmobiclip_hip <ver> gfx950 ..." */
const char *mobi_build_info(void);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
