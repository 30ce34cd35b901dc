// mobi_kernels.h -- launch interface between the C-ABI layer (mobi_abi.cpp) and mobi_kernels.hip.
#ifndef MOBI_KERNELS_H
#define MOBI_KERNELS_H
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "mobi_cmd.h"

// Everything one reconstruction launch needs.  HBM layout:
//   planes : [clip][slot 0..5][ Y: stride*height | UV: stride*height/2 ], each plane as macroblock TILES (mobi_tile.h): a
//            bijection of the reference's linear plane offsets (MD.cs:107-108,414-415), which the command list keeps using
//   desc / payload : the command list of one frame step, all clips (mobi_cmd.h)
// Ring: position r (0 = frame being written, 1..5 = references, MD.cs:102-106) lives in slot
//   (ring_base + 6 - r) % 6 ; every clip of a batch rotates in lock step, so ring_base is a scalar.
struct MobiReconArgs {
  uint8_t *planes;          // dwords 0-1
  const MbDesc *desc;       // 2-3   flat table of this frame step: [clip * n_mbs + mb]
  const uint32_t *payload;  // 4-5   payload arena of this frame step (MbDesc.payload_off indexes it)
  const int32_t *scale;     // 6-7   [quantizer][MOBI_SCALE_STRIDE] dequant scales by natural coefficient index
  int *fault;               // 8-9   [clip] clamp-table domain faults (MOBI_E_CLAMP)
  uint64_t clip_bytes;      // 10-11 6 * slot_bytes
  uint32_t slot_bytes;      // 12    stride*height*3/2
  int ring_base;            // 13
  int width, height;        // 14, 15
  int stride, mbw, n_mbs, n_clips;          // 16-19
  uint32_t pay_clip_words;                  // 20     0: MbDesc.payload_off indexes the whole arena; else it is relative to the clip's own
                                            //        part, payload + clip * pay_clip_words (device-parsed steps: arenas beyond 2^32 words)
  uint32_t reserved21;                      // 21
  uint32_t qpr, qpc, magic_qpr, magic_qpc;  // 22-25  octets (8 adjacent MBs = one wave) per MB row / per clip: filled in by mobi_launch_inter
  uint32_t step_tag;                        // 26     frame-step counter (never 0): done[] == step_tag means "reconstructed in this step"
  uint32_t inter_per_xcd;                   // 27     inter launch: workgroups per XCD (= gridDim.x / 8)
  uint32_t *done;                           // 28-29  [clip * n_mbs + mb] completion tags of intra macroblocks
  unsigned long long *prof;                 // 30-31  profiling accumulators (MOBI_DEBUG=9), else null
};
static_assert(sizeof(MobiReconArgs) == 128, "kernarg block layout");

// mobi_recon_inter8: every inter macroblock of the step, one wave per octet of macroblocks
extern "C" int mobi_launch_inter(const MobiReconArgs *a, hipStream_t s);
// items_dev: n_items launch items of 16 bytes (MOBI_INTRA_ITEM_WORDS words), sorted by dependency level -- see LevelPlan in mobi_abi.cpp
extern "C" int mobi_launch_intra(const MobiReconArgs *a, const uint32_t *items_dev, int n_items, hipStream_t s);
// small batches: both of the above in one launch (mobi_recon_step: the intra fours wait for the inter macroblocks their halo reads)
extern "C" int mobi_launch_step(const MobiReconArgs *a, const uint32_t *items_dev, int n_items, hipStream_t s);
// device-parsed frames: items_dev = [clip][n_mbs] in raster order, n_intra_dev[clip * stride_words] of them valid; K = the slots launched:
// the largest count when the host knows it (walk = 0), else MOBI_ASYNC_INTRA_SLOTS and walk = 1: a second launch, mobi_recon_intra_walk, one
// workgroup per four clips, goes through what the lists hold beyond K
#define MOBI_ASYNC_INTRA_SLOTS 160
extern "C" int mobi_launch_intra_cl(const MobiReconArgs *a, const uint32_t *items_dev, const uint32_t *n_intra_dev, int n_intra_stride_words, int K, int walk, hipStream_t s);
// MD.cs:260-323: ring slot 0 of clips [clip0, clip0 + n_clips) -> out_dev[clip][height][width] 0xAARRGGBB words (mobi_rgb.hip)
extern "C" int mobi_launch_argb(const MobiReconArgs *a, int version, int clip0, int n_clips, uint32_t *out_dev, hipStream_t s);
extern "C" long long mobi_launch_div239_check(hipStream_t s); // mismatches of the RGB kernel's x/239 over all floats, or -1
// Analyzer.cs:608-693: three-step 2x2 motion search of src_dev[clip][height][width] against ring slots 0..n_past-1 (mobi_analysis.hip)
extern "C" int mobi_launch_motion_search(const MobiReconArgs *a, const uint8_t *src_dev, uint32_t *out_dev, int n_past, hipStream_t s);
// MobiEncoder.DCT64 / DCT16 (Encoder/MobiEncoder.cs:962, 1146) of n_blocks residual blocks of n x n int32, n = 8 or 4 (mobi_analysis.hip)
extern "C" int mobi_launch_fwd_dct(int n, const int32_t *in_dev, int32_t *out_dev, uint32_t n_blocks, hipStream_t s);
// ring slot 0 of every clip against clip (clip mod modulus): out_dev[clip] += differing 16-byte words (mobi_analysis.hip); out_dev zeroed by the caller
extern "C" int mobi_launch_compare_clips(const MobiReconArgs *a, int modulus, uint32_t *out_dev, hipStream_t s);
// slot (tiled Y + UV planes of one frame) -> lin_dev: the same frame as the reference's row-major Y[stride*height] then UV[stride*height/2]
extern "C" int mobi_launch_untile(const uint8_t *slot, uint8_t *lin_dev, int stride, int height, hipStream_t s);
// intra launch item, word 0: (clip << 13) | mb; words 1..3: MbDesc.w1, MbDesc.payload_off, flags (mobi_recon_intra in mobi_kernels.hip)
#define MOBI_ITEM(clip, mb) (((uint32_t)(clip) << 13) | (uint32_t)(mb))
#define MOBI_INTRA_ITEM_WORDS 4
#define MOBI_ITEM_NONE 0xFFFFFFFFu /* padding: every dependency level starts on a wave of four items */
#endif
