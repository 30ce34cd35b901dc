// mobi_demux.cpp -- host-side container readers (include/mobiclip_demux.h): Mods and MOC5.
// Restated from LibMobiclip/Containers/Mods/ModsDemuxer.cs and MobiclipDecoder/Form1.cs:282-320; no GPU involved.
#include "../../include/mobiclip_demux.h"

#include <algorithm>
#include <cstring>
#include <new>
#include <vector>

namespace {
inline uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }                                     // IOUtil.ReadU16LE
inline uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); } // IOUtil.ReadU32LE
struct KeyFrame { uint32_t frame_number, data_offset; };
} // namespace

struct mobi_mods {
  const uint8_t *file;
  size_t len;
  mobi_mods_header h;
  std::vector<KeyFrame> key_frames;
  size_t pos;         // Stream.Position
  uint32_t cur_frame; // CurFrame
  int next_key_frame; // NextKeyFrame
};

extern "C" {

mobi_mods *mobi_mods_open(const uint8_t *file, size_t len) {
  if (!file || len < 0x30) return nullptr;
  mobi_mods *m = new (std::nothrow) mobi_mods();
  if (!m) return nullptr;
  m->file = file;
  m->len = len;
  mobi_mods_header &h = m->h; // ModsDemuxer.cs:48-64
  memcpy(h.mods_string, file, 4);
  h.tag_id = rd16(file + 4);
  h.tag_id_size_dword = rd16(file + 6);
  h.frame_count = rd32(file + 8);
  h.width = rd32(file + 0xC);
  h.height = rd32(file + 0x10);
  h.fps = rd32(file + 0x14);
  h.audio_codec = rd16(file + 0x18);
  h.nb_channel = rd16(file + 0x1A);
  h.frequency = rd32(file + 0x1C);
  h.biggest_frame = rd32(file + 0x20);
  h.audio_offset = rd32(file + 0x24);
  h.keyframe_index_offset = rd32(file + 0x28);
  h.keyframe_count = rd32(file + 0x2C);
  // audio codebooks (:21-29) and key frame index (:31-39) must lie inside the file
  if (h.audio_offset != 0 && (uint64_t)h.audio_offset + (uint64_t)h.nb_channel * MOBI_MODS_CODEBOOK_BYTES > len) { delete m; return nullptr; }
  if ((uint64_t)h.keyframe_index_offset + (uint64_t)h.keyframe_count * 8 > len) { delete m; return nullptr; }
  m->key_frames.resize(h.keyframe_count);
  for (uint32_t i = 0; i < h.keyframe_count; i++) {
    const uint8_t *p = file + h.keyframe_index_offset + (size_t)i * 8;
    m->key_frames[i] = KeyFrame{rd32(p), rd32(p + 4)};
  }
  m->pos = (size_t)h.keyframe_index_offset + (size_t)h.keyframe_count * 8; // where the constructor's reads leave the stream
  m->cur_frame = 0;
  m->next_key_frame = 0; // fields start at 0 in C#
  mobi_mods_jump_to_keyframe(m, 0);
  return m;
}
void mobi_mods_close(mobi_mods *m) { delete m; }
int mobi_mods_get_header(const mobi_mods *m, mobi_mods_header *out) {
  if (!m || !out) return -1;
  *out = m->h;
  return 0;
}
int mobi_mods_keyframe(const mobi_mods *m, int k, uint32_t *frame_number, uint32_t *data_offset) {
  if (!m || k < 0 || (size_t)k >= m->key_frames.size()) return -1;
  if (frame_number) *frame_number = m->key_frames[k].frame_number;
  if (data_offset) *data_offset = m->key_frames[k].data_offset;
  return 0;
}
const uint8_t *mobi_mods_audio_codebook(const mobi_mods *m, int channel) {
  if (!m || m->h.audio_offset == 0 || channel < 0 || channel >= m->h.nb_channel) return nullptr;
  return m->file + m->h.audio_offset + (size_t)channel * MOBI_MODS_CODEBOOK_BYTES;
}
void mobi_mods_jump_to_keyframe(mobi_mods *m, int k) { // :88-95
  if (!m || k < 0 || (uint32_t)k >= m->h.keyframe_count) return;
  m->pos = m->key_frames[k].data_offset;
  m->cur_frame = m->key_frames[k].frame_number;
  m->next_key_frame = (size_t)k + 1 < m->key_frames.size() ? k + 1 : -1;
}
int mobi_mods_read_frame(mobi_mods *m, const uint8_t **packet, uint32_t *packet_size, uint32_t *nr_audio_packets, int *is_key_frame) { // :97-116
  if (!m || !packet || !packet_size) return -1;
  if (nr_audio_packets) *nr_audio_packets = 0;
  if (is_key_frame) *is_key_frame = 0;
  if (m->cur_frame >= m->h.frame_count) return 0;
  if (m->next_key_frame >= 0 && (size_t)m->next_key_frame < m->key_frames.size() && m->cur_frame == m->key_frames[m->next_key_frame].frame_number) {
    if (is_key_frame) *is_key_frame = 1;
    m->next_key_frame = (size_t)m->next_key_frame + 1 < m->key_frames.size() ? m->next_key_frame + 1 : -1;
  }
  m->cur_frame++;
  if (m->pos + 4 > m->len) return -1;
  const uint32_t info = rd32(m->file + m->pos);
  m->pos += 4;
  const uint32_t size = info >> 14;
  if (nr_audio_packets) *nr_audio_packets = info & 0x3FFF;
  if (m->pos + size > m->len) return -1;
  *packet = m->file + m->pos;
  *packet_size = size;
  m->pos += size;
  return 1;
}

int mobi_moc5_open(const uint8_t *file, size_t len, mobi_moc5_info *info) { // Form1.cs:284-289
  if (!file || !info || len < 0x24) return -1;
  info->first_block = rd32(file + 4) + 8;
  info->width = rd32(file + 0x1C);
  info->height = rd32(file + 0x20);
  info->fps_x128 = rd32(file + 0xC);
  return 0;
}
int mobi_moc5_next_block(const uint8_t *file, size_t len, uint32_t *offs, int32_t *decode_offset, uint32_t *block_size) { // :293-318
  if (!file || !offs) return -1;
  if (*offs >= len) return 0;
  if ((size_t)*offs + 4 > len) return -1;
  const uint32_t bs = rd32(file + *offs);
  if (block_size) *block_size = bs;
  if (decode_offset) *decode_offset = (int32_t)(*offs + 8);
  // (64-bit: a block size near 2^32 must not wrap the offset backwards and make the caller loop for ever)
  const uint64_t o = ((uint64_t)*offs + 4 + (bs & ~1u) + 3) & ~(uint64_t)3;
  if (o <= *offs) return -1; // (the only hard error: the offset would not advance)
  // A last block that claims more than the file holds is still handed to the decoder, as the reference's loop does (Form1.cs:282-320:
  // it passes the whole file as Data and the decoder reads what is there); the following call reports the end of the file.
  const uint64_t next = std::min<uint64_t>(o, (uint64_t)len);
  if (next > 0xFFFFFFFFull || next <= *offs) return -1; // (files beyond 4 GB: the 32-bit offset cannot follow; never backwards -- ADVICE r03)
  *offs = (uint32_t)next;
  return 1;
}

} // extern "C"
