// mobi_moflex.cpp -- reader for the Moflex ("MoLive") container of 3DS Mobiclip files, include/mobiclip_demux.h.
//
// Written from the container grammar below; the reference's reader (LibMobiclip/Containers/Moflex/MoLiveDemux.cs:67-414, with
// MoLive.cs:34-51, MoLiveInBitStream.cs:16-54 and the MoLiveStream*.cs chunk classes) is the authority for the grammar, the
// return codes (callers stop on 73, Program.cs:164-166) and the corner cases named in the comments.
//
//   file    := packet*                       all packets have the size the last sync header announced (or the header-less
//                                            default window of 4096 bytes before the first one)
//   packet  := [sync] block ep* [pad]
//   sync    := 'L' '2'  check:u16be  time:u64be  (size-1):u16be  chunk*        14 bytes + chunks
//              check = the four 16-bit words of `time` XORed together, XOR 0xAAAA; bit 63 of `time` flags a reference
//              time stamp and does not take part in the check
//   chunk   := type:varint  size:varint  body[size]        type 0 ends the list (its body is skipped);
//              1 video (12 bytes) / 3 video with layout (13): stream u8, codec u8, fps rate, fps scale, width, height u16be,
//              pel ratio rate, scale u8 [, layout | rotation << 4 : u8]; 2 audio (6): stream, codec, (frequency - 1):u24be,
//              (channels - 1):u8; 4 timeline (2): stream, associated stream.  A stream chunk opens an endpoint for its stream index.
//   varint  := 1..4 bytes, 7 bits each, most significant first, bit 7 = "one more"; a fourth byte gives all its 8 bits
//   block   := flags:u8 [counter:u16be]      bit 0: packets vary in size; bit 1: a packet counter follows; bits 7..2: sync counter
//   ep      := header payload[size]          header, MSB first, padded to a byte: stream index as (unary length n, n bits);
//              end-of-frame bit; if set: frame type (unary length, bits), a time stamp (sign bit, unary length 28 + 2k, bits);
//              (size - 1):13 bits.  The payload is appended to the stream's frame; end-of-frame appends two zero bytes and
//              hands the frame out.
//   pad     := 0x00 ...                      a zero byte where an ep would start: the rest of a fixed-size packet is padding
#include "../../include/mobiclip_demux.h"

#include <cstring>
#include <deque>
#include <map>
#include <new>
#include <vector>

namespace {

struct Escape {}; // managed code would have thrown here (read past the byte[] it holds, duplicate dictionary key, Pop(> 64))

// What one ReadPacket() call can see: the packet array of the reference, filled from the reader position.  Indexing past the array is
// the reference's IndexOutOfRangeException.
// The reference reads into `new byte[PacketSize == 0 ? 0x1000 : PacketSize]` (MoLiveDemux.cs:71): bytes of that array behind what the
// file still held are zero, not out of range -- `cap` is the array's length, `n` the bytes actually read.
struct Window {
  const uint8_t *p = nullptr;
  size_t n = 0, cap = 0;
  uint8_t operator[](size_t i) const {
    if (i >= cap) throw Escape{};
    return i < n ? p[i] : 0;
  }
  uint32_t be16(size_t i) const { return (uint32_t)(*this)[i] << 8 | (*this)[i + 1]; }
  uint32_t be24(size_t i) const { return be16(i) << 8 | (*this)[i + 2]; }
  uint64_t be64(size_t i) const {
    uint64_t v = 0;
    for (int k = 0; k < 8; k++) v = v << 8 | (*this)[i + k];
    return v;
  }
};

// Bits, most significant first.  A byte is fetched only when a bit of it is needed, so `next_byte` is where byte-aligned
// data continues after a header (MoLiveInBitStream.Pop consumes exactly the same bytes for reads of up to 64 bits).
class BitReader {
 public:
  BitReader(const Window &w, size_t at) : w_(w), next_byte(at) {}
  uint64_t take(int n) {
    if (n > 64) throw Escape{}; // Pop: ArgumentException
    uint64_t v = 0;
    while (n > 0) {
      if (have_ == 0) { cur_ = w_[next_byte++]; have_ = 8; }
      const int k = n < have_ ? n : have_;
      v = v << k | ((uint64_t)(cur_ >> (have_ - k)) & ((1u << k) - 1u));
      have_ -= k;
      n -= k;
    }
    return v;
  }
  int unary() { // zeros before the first one bit
    int z = 0;
    while (take(1) == 0) z++;
    return z;
  }

 private:
  const Window &w_;
  uint32_t cur_ = 0;
  int have_ = 0;

 public:
  size_t next_byte;
};

struct SyncHeader {
  uint64_t time = 0;
  uint32_t packet_size = 0; // as the reference keeps it: a ushort (0xFFFF + 1 wraps to 0)
};
enum { kSyncBytes = 14, kDefaultWindow = 0x1000, kNoSyncCounter = 64, kNoPacketCounter = 65536 };

bool parse_sync(const Window &w, size_t at, SyncHeader &h) {
  if (w[at] != 'L' || w[at + 1] != '2') return false;
  const uint32_t check = w.be16(at + 2);
  const uint64_t t = w.be64(at + 4);
  h.time = t;
  h.packet_size = (w.be16(at + 12) + 1) & 0xFFFFu;
  // the reference-time flag (bit 63) stays out of the check word -- except for the one value 0x80000000_xxxxxxxx whose upper
  // half the reference's signed test ((int)(hi - 1) < 0) lets through unmasked
  const uint32_t hi = (uint32_t)(t >> 32), folded = ((hi - 1u) & 0x80000000u) ? hi & 0x7FFFFFFFu : hi;
  const uint32_t sum = (uint32_t)(t & 0xFFFF) ^ (uint32_t)(t >> 16 & 0xFFFF) ^ (folded & 0xFFFF) ^ (folded >> 16) ^ 0xAAAAu;
  return check == sum;
}
// the time stamp as the demuxer uses it: flag bit dropped (same signed quirk: 0x8000000000000000 itself keeps it)
uint64_t plain_time(uint64_t t) { return ((t - 1u) >> 63) ? t & 0x7FFFFFFFFFFFFFFFull : t; }

bool read_varint(const Window &w, uint32_t &value, uint32_t &pos, uint32_t limit) {
  value = 0;
  for (int k = 0; k < 4; k++) {
    if (pos == limit) return false;
    const uint8_t b = w[pos++];
    if (k == 3) { value |= b; return true; } // all 8 bits of the fourth byte, OR-ed onto the 21 bits already shifted up by 7 (MoLive.cs:49)
    if (!(b & 0x80)) { value |= b; return true; }
    value = (value | (b & 0x7Fu)) << 7;
  }
  return true;
}

struct Endpoint {
  mobi_moflex_stream info;
  std::vector<uint8_t> frame; // payloads of the frame being assembled
};

} // namespace

struct mobi_moflex {
  const uint8_t *file = nullptr;
  size_t len = 0, pos = 0;
  // demuxer state (MoLiveDemux.cs:22-33, :56-65)
  bool in_sync = false;
  uint32_t packet_size = 0;
  uint64_t last_time = 0, time_step = 0;       // Gts, DeltaGts
  uint32_t sync_counter = kNoSyncCounter, packet_counter = kNoPacketCounter;
  bool variable_size = false;
  std::map<int, Endpoint> streams;
  // frames handed out
  struct Frame { mobi_moflex_stream info; std::vector<uint8_t> data; };
  std::deque<Frame> ready;
  Frame current;

  void lose_sync() { // Desynchronize(), :56-65
    last_time = time_step = 0;
    sync_counter = kNoSyncCounter;
    packet_counter = kNoPacketCounter;
    in_sync = false;
    streams.clear();
  }

  // ---- stream table -------------------------------------------------------------------------------------------------
  // A chunk body is parsed against the WHOLE window, like chunk.Read(packet, pos): fields that do not fit stay 0 and the
  // chunk is accepted all the same (Read returns -1, the caller only rejects 0).
  static mobi_moflex_stream parse_stream(uint32_t type, const Window &w, size_t at) {
    mobi_moflex_stream s;
    std::memset(&s, 0, sizeof(s));
    s.chunk_id = type;
    s.stream_index = w[at];
    if (at + 1 >= w.cap) return s;
    if (type == 4) { s.associated_stream_index = w[at + 1]; return s; }
    s.codec_id = w[at + 1];
    const size_t body = at + 2, left = w.cap - body;
    if (type == 2) {
      if (left < 4) return s;
      s.frequency = w.be24(body) + 1;
      s.channel = (uint32_t)w[body + 3] + 1;
      return s;
    }
    if (left < 10) return s;
    s.fps_rate = w.be16(body);
    s.fps_scale = w.be16(body + 2);
    s.width = w.be16(body + 4);
    s.height = w.be16(body + 6);
    if (type == 1) {
      s.pel_ratio_rate = w[body + 8];
      s.pel_ratio_scale = w[body + 9];
      return s;
    }
    s.pel_ratio_rate = w[body + 9]; // MoLiveStreamVideoWithLayout.cs:40-41 stores both bytes into PelRatioRate: the scale stays 0
    if (body + 10 >= w.cap) return s;
    s.image_layout = w[body + 10] & 0xF;
    s.image_rotation = w[body + 10] >> 4;
    return s;
  }
  // one chunk of the list behind a sync header; 0x100 = the terminator was read
  uint32_t read_chunk(const Window &w, uint32_t &pos, uint32_t limit) {
    uint32_t type, size;
    if (!read_varint(w, type, pos, limit) || !read_varint(w, size, pos, limit)) { lose_sync(); return 0x43; }
    if (type == 0) { pos += size; return 0x100; }
    uint32_t want;
    switch (type) {
      case 1: want = 12; break;
      case 2: want = 6; break;
      case 3: want = 13; break;
      case 4: want = 2; break;
      case 0x100000: want = 20; break;
      default: return 0x44;
    }
    if (want != size) return 0x45;
    if (type == 0x100000) throw Escape{}; // MoLiveChunkFoo.Read: NotImplementedException
    const mobi_moflex_stream s = parse_stream(type, w, pos);
    if (streams.count(s.stream_index)) throw Escape{}; // Dictionary.Add: the key exists
    streams[s.stream_index].info = s;
    pos += size;
    if (pos <= limit) return 0;
    lose_sync();
    return 0x43;
  }

  // ---- data block ---------------------------------------------------------------------------------------------------
  uint32_t read_block(const Window &w, uint32_t &pos, uint32_t limit) {
    if (pos >= limit) { lose_sync(); return 67; }
    const uint8_t flags = w[pos++];
    variable_size = flags & 1;
    const bool counted = flags & 2;
    const uint32_t counter = flags >> 2;
    if (sync_counter == kNoSyncCounter) sync_counter = counter;
    else if (sync_counter != counter) {                 // sync headers were skipped: move the clock on, drop partial frames
      if (time_step == 0) { lose_sync(); return 70; }
      last_time += (uint64_t)(uint32_t)(counter - sync_counter) * time_step;
      sync_counter = counter;
      for (auto &kv : streams) kv.second.frame.clear();
    }
    if (counted) {
      const uint32_t seen = w.be16(pos);
      pos += 2;
      if (pos > limit) { lose_sync(); return 67; }
      if (packet_counter != kNoPacketCounter && packet_counter + 1 != seen) { packet_counter = kNoPacketCounter; return 0x50; }
      packet_counter = seen;
    }
    return 0;
  }

  // ---- elementary packets -------------------------------------------------------------------------------------------
  // 0 = one ep consumed and more may follow, 0x101 = the packet is finished
  uint32_t read_ep(const Window &w, uint32_t &pos, uint32_t limit) {
    if (pos == limit) return 0x101;
    if (pos > limit) { lose_sync(); return 0x43; }
    if (w[pos] == 0) { // padding
      pos++;
      if (!variable_size) pos = packet_size;
      return 0x101;
    }
    BitReader bits(w, pos);
    const int index_bits = bits.unary() + 1;
    const int stream = (int)bits.take(index_bits);
    const bool end_of_frame = bits.take(1) == 1;
    if (end_of_frame) {               // frame type and time stamp: read and dropped, as in the reference
      (void)bits.take(bits.unary() + 1);
      (void)bits.take(1);
      (void)bits.take(28 + 2 * bits.unary());
    }
    const uint32_t size = (uint32_t)bits.take(13) + 1;
    pos = (uint32_t)bits.next_byte;
    if (pos + size > limit) { lose_sync(); return 0x43; }
    auto it = streams.find(stream);
    if (it != streams.end()) it->second.frame.insert(it->second.frame.end(), w.p + pos, w.p + pos + size);
    pos += size;
    if (end_of_frame && it != streams.end()) {
      Frame f;
      f.info = it->second.info;
      f.data.swap(it->second.frame);
      f.data.push_back(0); // what the decoder's 16-bit read-ahead may touch behind the last code (MoLiveDemux.cs:353)
      f.data.push_back(0);
      ready.push_back(std::move(f));
    }
    return pos < limit ? 0 : 0x101;
  }

  // ---- one packet ---------------------------------------------------------------------------------------------------
  uint32_t read_packet() {
    Window w;
    const size_t want = packet_size ? packet_size : kDefaultWindow;
    w.p = file + pos;
    w.n = pos < len ? (want < len - pos ? want : len - pos) : 0;
    w.cap = want;
    const uint32_t length = (uint32_t)w.n;
    SyncHeader h;
    if (!in_sync) { // look for a header whose check word fits
      if (length < kSyncBytes) return 1;
      uint32_t at = 0;
      while (!parse_sync(w, at, h)) {
        at++;
        if (at == length - kSyncBytes) return 0x80; // none in this window
      }
      if (h.packet_size < 0x10) return 73;
      in_sync = true;
      pos += at;
      return 0;
    }
    if (packet_size != 0 && packet_size != length) return 73; // a short last packet: the end of the file for the callers
    uint32_t cur = 0;
    if (length > kSyncBytes && parse_sync(w, 0, h)) {
      if (h.packet_size < 0x10) return 73;
      const uint64_t t = plain_time(h.time);
      if (t != 0) {
        if (last_time != 0 && time_step == 0) time_step = t - last_time;
        last_time = t;
        streams.clear(); // the chunks that follow list the streams again
      }
      if (packet_size != h.packet_size) {
        const bool window_too_small = want < h.packet_size;
        packet_size = h.packet_size;
        if (window_too_small) return 0; // look at this packet again, whole
      }
      cur = kSyncBytes;
      const uint32_t limit = packet_size > length ? length : packet_size;
      for (;;) {
        const uint32_t r = read_chunk(w, cur, limit);
        if (r == 0x100) break;
        if (r != 0) return r;
      }
      if (cur > length) return 0x43;
    }
    uint32_t r = read_block(w, cur, length);
    if (!in_sync) return 0;
    if (r != 0) return r;
    for (;;) {
      r = read_ep(w, cur, length);
      if (r == 0x101) break;
      if (r != 0) return r;
    }
    if (cur > length) return 0x43;
    pos += cur;
    return 0;
  }
};

extern "C" {

mobi_moflex *mobi_moflex_open(const uint8_t *file, size_t len) {
  if (!file) return nullptr;
  mobi_moflex *m = new (std::nothrow) mobi_moflex();
  if (!m) return nullptr;
  m->file = file;
  m->len = len;
  return m;
}
void mobi_moflex_close(mobi_moflex *m) { delete m; }

int mobi_moflex_read_packet(mobi_moflex *m) {
  if (!m) return -1;
  try {
    return (int)m->read_packet();
  } catch (const Escape &) {
    return -1;
  } catch (const std::bad_alloc &) {
    return -1;
  }
}
int mobi_moflex_pop_frame(mobi_moflex *m, mobi_moflex_stream *stream, const uint8_t **data, size_t *len) {
  if (!m || m->ready.empty()) return 0;
  m->current = std::move(m->ready.front());
  m->ready.pop_front();
  if (stream) *stream = m->current.info;
  if (data) *data = m->current.data.data();
  if (len) *len = m->current.data.size();
  return 1;
}
int mobi_moflex_next_frame(mobi_moflex *m, mobi_moflex_stream *stream, const uint8_t **data, size_t *len) {
  if (!m) return -1;
  int idle = 0; // calls in a row that neither moved the reader nor produced a frame
  for (;;) {
    if (mobi_moflex_pop_frame(m, stream, data, len)) return 1;
    const size_t before = m->pos;
    const uint32_t before_size = m->packet_size;
    const int rc = mobi_moflex_read_packet(m);
    if (rc == 73 || rc == 1) return 0; // a short last packet, or fewer than 14 bytes left to search: the stream is over
    if (rc != 0) return rc < 0 ? rc : -rc;
    // A damaged packet can make the reader lose and regain synchronisation on the same header for ever (ReadPacket returns 0
    // both times and the position never moves; the reference's callers would spin).  Three idle calls cover the legitimate
    // cases (sync found at offset 0, then the window enlarged, then the packet read); after that: give up.
    if (m->pos == before && m->packet_size == before_size && m->ready.empty()) {
      if (++idle >= 3) return -0x43;
    } else idle = 0;
  }
}

} // extern "C"
