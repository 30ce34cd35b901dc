/*
 * mobiclip_demux.h -- C ABI of the host-side container readers that hand frames to the decoder.
 *
 *   Mods (.mods, Nintendo DS)   mirrors LibMobiclip.Containers.Mods.ModsDemuxer (ModsDemuxer.cs:16-117)
 *   MOC5 (Wii)                  mirrors the inline parser of the reference GUI (MobiclipDecoder/Form1.cs:282-320)
 *   Moflex (.moflex, 3DS)       mirrors LibMobiclip.Containers.Moflex.MoLiveDemux (MoLiveDemux.cs)
 *
 * Pure byte parsing on the host: no GPU, no copies -- every pointer handed out points into the caller's file buffer,
 * which must outlive the handle.  Where the reference would throw (reads past the end of the stream) these return
 * an error instead.
 */
#ifndef MOBICLIP_DEMUX_H
#define MOBICLIP_DEMUX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default) /* the library is built with -fvisibility=hidden: these declarations are all it exports */
#endif

/* ---- Mods --------------------------------------------------------------------------------------- */
typedef struct mobi_mods mobi_mods;

/* ModsDemuxer.ModsHeader (ModsDemuxer.cs:44-79): the first 0x30 bytes of the file, little endian */
typedef struct {
  char mods_string[4];          /* 0x00 */
  uint16_t tag_id;              /* 0x04 */
  uint16_t tag_id_size_dword;   /* 0x06 */
  uint32_t frame_count;         /* 0x08 */
  uint32_t width, height;       /* 0x0C, 0x10 */
  uint32_t fps;                 /* 0x14  (raw field, as the reference keeps it) */
  uint16_t audio_codec;         /* 0x18 */
  uint16_t nb_channel;          /* 0x1A */
  uint32_t frequency;           /* 0x1C */
  uint32_t biggest_frame;       /* 0x20 */
  uint32_t audio_offset;        /* 0x24 */
  uint32_t keyframe_index_offset; /* 0x28 */
  uint32_t keyframe_count;      /* 0x2C */
} mobi_mods_header;

#define MOBI_MODS_CODEBOOK_BYTES 0xC34 /* ModsDemuxer.cs:26 */

/* new ModsDemuxer(stream) (:16-42): header, audio codebooks, key frame index, JumpToKeyFrame(0).
 * NULL when the file is shorter than what the header points at. */
mobi_mods *mobi_mods_open(const uint8_t *file, size_t len);
void mobi_mods_close(mobi_mods *m);
int mobi_mods_get_header(const mobi_mods *m, mobi_mods_header *out);
/* KeyFrames[k] (:81-86); returns 0, or -1 when k is out of range */
int mobi_mods_keyframe(const mobi_mods *m, int k, uint32_t *frame_number, uint32_t *data_offset);
/* AudioCodebooks[channel] (:21-29): MOBI_MODS_CODEBOOK_BYTES bytes, or NULL (no audio / bad channel) */
const uint8_t *mobi_mods_audio_codebook(const mobi_mods *m, int channel);
/* JumpToKeyFrame(k) (:88-95): silently ignored when k >= KeyframeCount, like the reference */
void mobi_mods_jump_to_keyframe(mobi_mods *m, int k);
/* ReadFrame(out NrAudioPackets, out IsKeyFrame) (:97-116).  Returns 1 and the packet (video bits first, then the audio
 * packets; feed it to mobi_decode with offset 0), 0 when CurFrame >= FrameCount (the reference returns null),
 * -1 when the file ends inside the packet. */
int mobi_mods_read_frame(mobi_mods *m, const uint8_t **packet, uint32_t *packet_size, uint32_t *nr_audio_packets, int *is_key_frame);

/* ---- MOC5 --------------------------------------------------------------------------------------- */
typedef struct {
  uint32_t width, height;   /* u32 at 0x1C, 0x20 (Form1.cs:286-287) */
  uint32_t fps_x128;        /* u32 at 0x0C; frames per second = fps_x128 / 128 (:289) */
  uint32_t first_block;     /* u32 at 0x04, + 8 (:285) */
} mobi_moc5_info;
/* returns 0, or -1 when the file is too short */
int mobi_moc5_open(const uint8_t *file, size_t len, mobi_moc5_info *info);
/* One iteration of the frame loop (:293-318): the block at *offs.  Returns 1 and decode_offset = *offs + 8 (the
 * decoder gets the WHOLE file as Data and this Offset), then advances *offs by 4 + (blocksize & ~1), rounded up to a
 * multiple of 4 (never past len); returns 0 when *offs >= len (the reference exits).  A last block whose size field claims more
 * than the file holds is still returned (1): the reference's loop hands it to the decoder too, which reads what is there.  -1 only
 * for bad arguments, fewer than 4 bytes left for the size field, or a size that would not advance the offset (2^32 wrap). */
int mobi_moc5_next_block(const uint8_t *file, size_t len, uint32_t *offs, int32_t *decode_offset, uint32_t *block_size);

/* ---- Moflex (3DS) ------------------------------------------------------------------------------ */
/* A reader for the container LibMobiclip.Containers.Moflex.MoLiveDemux reads (MoLiveDemux.cs:11-416), with the file held in
 * memory: packets of the announced size, a sync header with a check word and the stream table, a flags byte per packet,
 * bit-packed elementary-packet headers, per-stream frame assembly (grammar at the top of mobi_moflex.cpp).  It keeps the
 * reference's ReadPacket() return codes; completed frames -- what the reference hands to OnCompleteFrameReceived, two zero
 * bytes appended (:353) -- are queued. */
typedef struct mobi_moflex mobi_moflex;

/* MoLiveStream chunk of a frame (MoLiveStreamVideo.cs / ...WithLayout.cs / ...Audio.cs / ...Timeline.cs) */
typedef struct {
  uint32_t chunk_id;       /* 1 video, 2 audio, 3 video with layout, 4 timeline (MoLiveDemux.cs:181-199) */
  int32_t stream_index;
  uint32_t codec_id;
  uint32_t fps_rate, fps_scale, width, height, pel_ratio_rate, pel_ratio_scale; /* video */
  uint32_t image_layout, image_rotation;                                         /* video with layout */
  uint32_t frequency, channel;                                                   /* audio */
  uint32_t associated_stream_index;                                              /* timeline */
} mobi_moflex_stream;

mobi_moflex *mobi_moflex_open(const uint8_t *file, size_t len);
void mobi_moflex_close(mobi_moflex *m);
/* MoLiveDemux.ReadPacket() (:67-160): returns the reference's code (0 ok; 73 = end of stream / packet size mismatch, the
 * value the callers stop on, Program.cs:164-166; 1, 0x43.., 0x80 as in the source); -1 where the reference would throw. */
int mobi_moflex_read_packet(mobi_moflex *m);
/* Pops the oldest completed frame; the data pointer stays valid until the next call on this handle.  1 = a frame, 0 = none. */
int mobi_moflex_pop_frame(mobi_moflex *m, mobi_moflex_stream *stream, const uint8_t **data, size_t *len);
/* Convenience: ReadPacket() until a frame is available.  1 = a frame, 0 = the stream ended (code 73, or code 1: fewer than 14
 * bytes left), <0 = demux error (the negated ReadPacket code, or -1; -0x43 also when a damaged packet makes the reader lose and
 * regain synchronisation on the same bytes without ever advancing -- the reference's callers would spin there). */
int mobi_moflex_next_frame(mobi_moflex *m, mobi_moflex_stream *stream, const uint8_t **data, size_t *len);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
