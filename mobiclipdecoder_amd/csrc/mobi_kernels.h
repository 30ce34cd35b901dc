// mobi_kernels.h -- launch interface between the C-ABI layer (mobi_abi.cpp) and mobi_kernels.hip.
#ifndef MOBI_KERNELS_H
#define MOBI_KERNELS_H
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "mobi_cmd.h"

// Everything one reconstruction launch needs.  HBM layout:
//   planes : [clip][slot 0..5][ Y: stride*height | UV: stride*height/2 ]   (the reference's own plane
//            layout, MD.cs:107-108,414-415, so linear offsets in the command list apply unchanged)
//   desc / payload : the command list of one frame step, all clips (mobi_cmd.h)
// Ring: position r (0 = frame being written, 1..5 = references, MD.cs:102-106) lives in slot
//   (ring_base + 6 - r) % 6 ; every clip of a batch rotates in lock step, so ring_base is a scalar.
struct MobiReconArgs {
  uint8_t *planes;
  const MbDesc *desc;       // flat table of this frame step: [clip * n_mbs + mb]
  const uint32_t *payload;  // payload arena of this frame step (MbDesc.payload_off indexes it)
  const int32_t *scale;     // [quantizer][MOBI_SCALE_STRIDE] dequant scales by natural coefficient index
  int *fault;               // [clip] clamp-table domain faults (MOBI_E_CLAMP)
  uint64_t clip_bytes;      // 6 * slot_bytes
  uint32_t slot_bytes;      // stride*height*3/2
  int ring_base;
  int width, height, stride, mbw, n_mbs, n_clips;
  uint32_t magic_n_mbs, magic_mbw; // floor(2^32 / d) for the in-kernel divisions (no 64-bit divides on the GPU)
  uint32_t opr, opc, magic_opr, magic_opc; // octets (8 MBs) per MB row / per clip, and their magics
  int debug;                       // profiling aid (env MOBI_DEBUG): 0 = normal
};

extern "C" int mobi_launch_inter(const MobiReconArgs *a, hipStream_t s);
extern "C" int mobi_launch_intra(const MobiReconArgs *a, const uint32_t *items_dev, int n_items, hipStream_t s);
// intra launch item: (clip << 13) | mb
#define MOBI_ITEM(clip, mb) (((uint32_t)(clip) << 13) | (uint32_t)(mb))
#endif
