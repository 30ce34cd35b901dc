// mobi_moflex.cpp -- Moflex (MoLive) demuxer, include/mobiclip_demux.h.
// Restated from LibMobiclip/Containers/Moflex/MoLiveDemux.cs (ReadPacket :67, ReadSynchroChunk :164, ReadDataBlock :216,
// ReadEp :266, ReadSynchroHeader :377), MoLive.cs (ReadVariableByte :38), MoLiveInBitStream.cs (Pop :18) and the stream
// chunk readers (MoLiveStreamVideo.cs :33, ...WithLayout.cs :27, ...Audio.cs :16, ...Timeline.cs :14).  The file in memory
// plays the Stream: Read() copies what is left, Position moves as in the source.  C# semantics kept on purpose: shift
// counts are masked (& 63 for 64-bit, & 31 for 32-bit operands), uint arithmetic wraps, Dictionary.Add on an existing
// key and reads past an array are exceptions (-> return -1).
#include "../../include/mobiclip_demux.h"

#include <cstring>
#include <deque>
#include <map>
#include <new>
#include <vector>

namespace {
struct Thrown {}; // "the reference would have thrown here"
inline uint64_t shl64(uint64_t v, int n) { return v << (n & 63); }
inline uint64_t shr64(uint64_t v, int n) { return v >> (n & 63); }

struct Bytes { // byte[] with bounds checks
  const uint8_t *p; size_t n;
  uint8_t at(size_t i) const { if (i >= n) throw Thrown{}; return p[i]; }
};
inline uint32_t u16be(const Bytes &b, size_t o) { return ((uint32_t)b.at(o) << 8) | b.at(o + 1); }
inline uint32_t u24be(const Bytes &b, size_t o) { return ((uint32_t)b.at(o) << 16) | ((uint32_t)b.at(o + 1) << 8) | b.at(o + 2); }
inline uint32_t u32be(const Bytes &b, size_t o) { return (u16be(b, o) << 16) | u16be(b, o + 2); }

struct BitStream { // MoLiveInBitStream.cs:10-56
  uint64_t Value = 0; uint32_t Remaining = 0; Bytes Stream{nullptr, 0}; uint32_t Pos = 0;
  uint64_t Pop(int NrBits) {
    if (NrBits > 64) throw Thrown{};
    const uint32_t v5 = 64 - ((64 - Remaining) & 7);
    if ((uint32_t)NrBits > v5) {
      const uint32_t v12 = (uint32_t)NrBits - v5;
      if (Remaining < v5) {
        do { Value |= shl64((uint64_t)Stream.at(Pos++), (int)(56 - Remaining)); Remaining += 8; } while (Remaining < v5);
      }
      const uint8_t data = Stream.at(Pos++);
      const uint64_t res1 = shl64(shr64(Value, (int)(64 - v5)), (int)v12);
      Value = shl64((uint64_t)data, (int)(v12 + 56));
      Remaining = 8 - v12;
      return res1 | shr64((uint64_t)data, (int)(8 - v12));
    }
    if (Remaining < (uint32_t)NrBits) {
      do { Value |= shl64((uint64_t)Stream.at(Pos++), (int)(56 - Remaining)); Remaining += 8; } while (Remaining < (uint32_t)NrBits);
    }
    const uint64_t v10 = shr64(Value, 64 - NrBits);
    Value = shl64(Value, NrBits);
    Remaining -= (uint32_t)NrBits;
    return v10;
  }
};

bool ReadVariableByte(const Bytes &src, uint32_t &value, uint32_t &pos, uint32_t psize) { // MoLive.cs:38-55
  value = 0;
  if (pos == psize) return false;
  uint8_t data = src.at(pos++);
  if ((data & 0x80) == 0) { value = data; return true; }
  if (pos == psize) return false;
  value = (uint32_t)(data & 0x7F) << 7;
  data = src.at(pos++);
  if ((data & 0x80) == 0) { value |= data; return true; }
  if (pos == psize) return false;
  value = ((uint32_t)(data & 0x7F) | value) << 7;
  data = src.at(pos++);
  if ((data & 0x80) == 0) { value |= data; return true; }
  if (pos == psize) return false;
  value = (((uint32_t)(data & 0x7F) | value) << 7) | src.at(pos++);
  return true;
}

bool ReadSynchroHeader(const Bytes &packet, int offset, uint64_t &ts, uint16_t &packetSize) { // MoLiveDemux.cs:377-414
  ts = 0;
  packetSize = 0;
  if (!(packet.at(offset) == 0x4C && packet.at(offset + 1) == 0x32)) return false;
  offset += 2;
  const uint32_t v10 = u16be(packet, offset);
  offset += 2;
  const uint32_t v13 = (u32be(packet, offset) & 0xFFFFFF00u) | packet.at(offset + 3);
  offset += 4;
  const uint32_t v12 = (uint32_t)packet.at(offset++) << 24;
  const uint32_t v14 = packet.at(offset++);
  const uint32_t v15 = v12 | (v14 << 16);
  const uint32_t v16 = v13 | (v14 >> 16);
  const uint32_t v17 = packet.at(offset++);
  ts = (uint64_t)(v15 | (v17 << 8) | (uint32_t)packet.at(offset++)) | ((uint64_t)(v16 | (v17 >> 24)) << 32);
  uint32_t v19 = (uint32_t)(ts >> 32);
  if ((int32_t)(uint32_t)((ts >> 32) - 1) < 0) v19 &= 0x7FFFFFFFu;
  packetSize = (uint16_t)(u16be(packet, offset) + 1);
  return v10 == (uint32_t)(((ts >> 16) & 0xFFFF) ^ (v19 >> 16) ^ 0xAAAA ^ (v19 & 0xFFFF) ^ (ts & 0xFFFF));
}

struct Endpoint { mobi_moflex_stream chunk; std::vector<uint8_t> data; };
struct Frame { mobi_moflex_stream chunk; std::vector<uint8_t> data; };
} // namespace

struct mobi_moflex {
  const uint8_t *file = nullptr; size_t len = 0; size_t position = 0; // Reader
  uint64_t Gts = 0, DeltaGts = 0;
  uint32_t PacketSize = 0, SynchroCounter = 64, LastCounter = 0;
  bool VariablePacketSize = true, HasReferenceTs = false, Synchronized = false, ReaderIsDatagramBased = false;
  std::map<int, Endpoint> Streams;
  std::deque<Frame> done;
  Frame current; // what mobi_moflex_pop_frame handed out last

  void Desynchronize() { // :54-65
    Gts = 0; DeltaGts = 0; SynchroCounter = 64; LastCounter = 65536; Synchronized = false; Streams.clear();
  }

  uint32_t ReadSynchroChunk(const Bytes &packet, uint32_t &pos, uint32_t psize) { // :164-214
    uint32_t type, size;
    if (!ReadVariableByte(packet, type, pos, psize) || !ReadVariableByte(packet, size, pos, psize)) { Desynchronize(); return 0x43; }
    mobi_moflex_stream c;
    memset(&c, 0, sizeof(c));
    c.stream_index = -1;
    uint32_t chunk_size;
    switch (type) {
      case 0: pos += size; return 0x100;
      case 1: c.chunk_id = 1; chunk_size = 12; break;
      case 2: c.chunk_id = 2; chunk_size = 6; break;
      case 3: c.chunk_id = 3; chunk_size = 13; break;
      case 4: c.chunk_id = 4; chunk_size = 2; break;
      case 0x100000: throw Thrown{}; // MoLiveChunkFoo.Read: NotImplementedException
      default: return 0x44;
    }
    if (chunk_size != size) return 0x45;
    // chunk.Read(packet, pos): -1 (not an error for the caller, which only tests == 0) or the end offset
    int offset = (int)pos;
    auto rd = [&]() -> int {
      if (packet.n == 0) return -1;
      c.stream_index = packet.at(offset++);
      if ((size_t)offset >= packet.n) return -1;
      if (type == 4) { c.associated_stream_index = packet.at(offset++); return offset; }
      c.codec_id = packet.at(offset++);
      if (type == 2) {
        if ((long)packet.n - offset < 0x4) return -1;
        c.frequency = u24be(packet, offset) + 1;
        c.channel = (uint32_t)packet.at(offset + 3) + 1;
        return offset + 4;
      }
      if ((long)packet.n - offset < 0xA) return -1;
      c.fps_rate = u16be(packet, offset);
      c.fps_scale = u16be(packet, offset + 2);
      c.width = u16be(packet, offset + 4);
      c.height = u16be(packet, offset + 6);
      c.pel_ratio_rate = packet.at(offset + 8);
      c.pel_ratio_scale = packet.at(offset + 9);
      if (type == 3) c.pel_ratio_rate = packet.at(offset + 9); // MoLiveStreamVideoWithLayout.cs:40-41 assigns PelRatioRate twice; PelRatioScale stays 0
      if (type == 3) c.pel_ratio_scale = 0;
      offset += 0xA;
      if (type == 3) {
        if ((size_t)offset >= packet.n) return -1;
        c.image_layout = packet.at(offset) & 0xF;
        c.image_rotation = packet.at(offset) >> 4;
        offset++;
      }
      return offset;
    };
    if (rd() == 0) return 0x45;
    if (Streams.count(c.stream_index)) throw Thrown{}; // Dictionary.Add: ArgumentException
    Streams[c.stream_index] = Endpoint{c, {}};
    pos += size;
    if (pos <= psize) return 0;
    Desynchronize();
    return 0x43;
  }

  uint32_t ReadDataBlock(const Bytes &packet, uint32_t &pos, uint32_t psize) { // :216-259
    if (pos >= psize) { Desynchronize(); return 67; }
    const uint8_t flags = packet.at(pos++);
    VariablePacketSize = (flags & 1) == 1;
    const bool PacketCounting = ((flags >> 1) & 1) == 1;
    const uint32_t synchrocounter = (uint32_t)(flags >> 2);
    if (SynchroCounter == 64) SynchroCounter = synchrocounter;
    else if (SynchroCounter != synchrocounter) {
      if (DeltaGts == 0) { Desynchronize(); return 70; }
      Gts += (uint64_t)(synchrocounter - SynchroCounter) * DeltaGts;
      SynchroCounter = synchrocounter;
      for (auto &kv : Streams) kv.second.data.clear();
    }
    if (PacketCounting) {
      const uint32_t val = u16be(packet, pos);
      pos += 2;
      if (pos > psize) { Desynchronize(); return 67; }
      const uint32_t expectedval = LastCounter == 65536 ? val : LastCounter + 1;
      if (expectedval != val) { LastCounter = 65536; return 0x50; }
      LastCounter = val;
    }
    return 0;
  }

  uint32_t ReadEp(const Bytes &packet, uint32_t &pos, uint32_t psize) { // :266-375
    if (pos == psize) return 0x101;
    if (pos > psize) { Desynchronize(); return 0x43; }
    const uint8_t tmp = packet.at(pos);
    if (tmp == 0) {
      pos++;
      if (!VariablePacketSize) pos = PacketSize;
      return 0x101;
    }
    int NrStreamIdxBits = 1;
    BitStream bs;
    bs.Stream = packet;
    bs.Pos = pos;
    while (bs.Pop(1) == 0) NrStreamIdxBits++;
    const int StreamIdx = (int)bs.Pop(NrStreamIdxBits);
    const bool EndFrame = bs.Pop(1) == 1;
    if (EndFrame) { // frame type and time stamp delta: parsed, not used (:303-318)
      int FrameTypeNrBits = 1;
      while (bs.Pop(1) == 0) FrameTypeNrBits++;
      (void)bs.Pop(FrameTypeNrBits);
      int v23 = 28;
      (void)bs.Pop(1);
      while (bs.Pop(1) == 0) v23 += 2;
      (void)bs.Pop(v23);
    }
    const int EPSize = (int)bs.Pop(0xD) + 1;
    pos = bs.Pos;
    if (pos + (uint32_t)EPSize > psize) { Desynchronize(); return 0x43; }
    if ((size_t)pos + (size_t)EPSize > packet.n) throw Thrown{}; // Array.Copy past the buffer
    auto it = Streams.find(StreamIdx);
    if (it != Streams.end()) it->second.data.insert(it->second.data.end(), packet.p + pos, packet.p + pos + EPSize);
    pos += (uint32_t)EPSize;
    if (EndFrame && it != Streams.end()) {
      it->second.data.push_back(0); // AddData(new byte[2]), :353
      it->second.data.push_back(0);
      done.push_back(Frame{it->second.chunk, it->second.data}); // OnCompleteFrameReceived(Chunk, GetData())
      it->second.data.clear();
    }
    if (pos < psize) return 0;
    return 0x101;
  }

  uint32_t ReadPacket() { // :67-160
    uint64_t ts;
    uint16_t packetsize;
    const size_t want = PacketSize == 0 ? 0x1000 : PacketSize;
    std::vector<uint8_t> buf(want, 0);
    const size_t avail = position < len ? len - position : 0;
    const int length = (int)(avail < want ? avail : want); // Reader.Read; Position is put back right away
    if (length) memcpy(buf.data(), file + position, (size_t)length);
    const Bytes packet{buf.data(), buf.size()};
    if (!Synchronized) {
      if (length < 0xE) return 1;
      int offset = 0;
      while (!ReadSynchroHeader(packet, offset, ts, packetsize)) {
        offset++;
        if (offset == length - 0xE) return 0x80;
      }
      if ((int64_t)ts - 1 < 0) { HasReferenceTs = true; ts &= 0x7FFFFFFFFFFFFFFFull; } else HasReferenceTs = false;
      if (packetsize < 0x10) return 73;
      Synchronized = true;
      position += (size_t)offset;
      return 0;
    }
    if (!ReaderIsDatagramBased && PacketSize != 0 && PacketSize != (uint32_t)length) return 73;
    uint32_t offset2 = 0;
    if (length > 0xE && ReadSynchroHeader(packet, 0, ts, packetsize)) {
      if ((int64_t)ts - 1 < 0) { HasReferenceTs = true; ts &= 0x7FFFFFFFFFFFFFFFull; } else HasReferenceTs = false;
      if (packetsize < 0x10) return 73;
      if (ts != 0) {
        if (Gts != 0 && DeltaGts == 0) DeltaGts = ts - Gts;
        Gts = ts;
        Streams.clear();
      }
      if (PacketSize != packetsize) {
        const bool retry = (PacketSize == 0 ? 0x1000u : PacketSize) < packetsize;
        PacketSize = packetsize;
        if (retry) return 0;
      }
      offset2 = 0xE;
      const uint32_t size = PacketSize > (uint32_t)length ? (uint32_t)length : PacketSize;
      for (;;) {
        const uint32_t result = ReadSynchroChunk(packet, offset2, size);
        if (result == 0x100) break;
        if (result != 0) return result;
      }
      if (offset2 > (uint32_t)length) return 0x43;
    }
    uint32_t result2 = ReadDataBlock(packet, offset2, (uint32_t)length);
    if (!Synchronized) return 0;
    if (result2 == 0) {
      for (;;) {
        result2 = ReadEp(packet, offset2, (uint32_t)length);
        if (result2 == 0x101) break;
        if (result2 != 0) return result2;
      }
      if (offset2 > (uint32_t)length) return 0x43;
      position += offset2;
      return 0;
    }
    return result2;
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
  try { return (int)m->ReadPacket(); } catch (const Thrown &) { return -1; } catch (const std::bad_alloc &) { return -1; }
}
int mobi_moflex_pop_frame(mobi_moflex *m, mobi_moflex_stream *stream, const uint8_t **data, size_t *len) {
  if (!m || m->done.empty()) return 0;
  m->current = std::move(m->done.front());
  m->done.pop_front();
  if (stream) *stream = m->current.chunk;
  if (data) *data = m->current.data.data();
  if (len) *len = m->current.data.size();
  return 1;
}
int mobi_moflex_next_frame(mobi_moflex *m, mobi_moflex_stream *stream, const uint8_t **data, size_t *len) {
  if (!m) return -1;
  for (;;) {
    if (mobi_moflex_pop_frame(m, stream, data, len)) return 1;
    const int rc = mobi_moflex_read_packet(m);
    if (rc == 73) return m->done.empty() ? 0 : mobi_moflex_pop_frame(m, stream, data, len);
    if (rc < 0) return -1;
    if (rc != 0 && rc != 1 && rc != 0x50) return -rc; // 1: too little data yet / 0x50: packet counter gap -- the callers just keep reading
    if (rc == 1) return m->done.empty() ? 0 : mobi_moflex_pop_frame(m, stream, data, len); // fewer than 14 bytes left before synchronisation: nothing more will come
  }
}

} // extern "C"
