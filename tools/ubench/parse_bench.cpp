// Host parser micro-benchmark (CPU only): frames per second of MobiStreamParser::parse_frame on a generated clip.
//   g++ -O2 -std=c++17 -I mobiclipdecoder_amd/csrc tools/ubench/parse_bench.cpp mobiclipdecoder_amd/csrc/mobi_parse.cpp -o /tmp/parse_bench
//   python -c "..." writes the clip (see tools/ubench/README or run via tools/exp_parse_cpu.py)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mobi_parse.h"

int main(int argc, char **argv) {
  if (argc < 5) { fprintf(stderr, "usage: parse_bench clip.bin offsets.bin width height [version] [reps]\n"); return 2; }
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 2;
  std::vector<uint8_t> data;
  fseek(f, 0, SEEK_END);
  data.resize((size_t)ftell(f));
  fseek(f, 0, SEEK_SET);
  if (fread(data.data(), 1, data.size(), f) != data.size()) return 2;
  fclose(f);
  f = fopen(argv[2], "rb");
  if (!f) return 2;
  std::vector<uint32_t> fo;
  uint32_t v;
  while (fread(&v, 4, 1, f) == 1) fo.push_back(v);
  fclose(f);
  const int w = atoi(argv[3]), h = atoi(argv[4]), ver = argc > 5 ? atoi(argv[5]) : 2, reps = argc > 6 ? atoi(argv[6]) : 20;
  const int nf = (int)fo.size() - 1;
  ParsedFrame pf;
  double best = 1e30;
  size_t cmd = 0;
  for (int r = 0; r < reps; r++) {
    MobiStreamParser p((uint32_t)w, (uint32_t)h, ver);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < nf; i++) {
      int32_t off = (int32_t)fo[i];
      int rc = p.parse_frame(data.data(), fo[i + 1], &off, pf);
      if (rc != 0) { fprintf(stderr, "frame %d rc %d\n", i, rc); return 1; }
      cmd += pf.cmd_bytes();
    }
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (dt < best) best = dt;
  }
  printf("%d frames %dx%d: %.3f ms per frame (best of %d), %.1f Mpix/s per core, %.1f KB of commands per frame\n", nf, w, h, best * 1e3 / nf, reps,
         (double)w * h * nf / best / 1e6, (double)cmd / reps / nf / 1024);
  return 0;
}
