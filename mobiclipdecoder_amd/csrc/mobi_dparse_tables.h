// mobi_dparse_tables.h -- layout of the table blob of the device-side parser (no HIP types: mobi_parse.cpp builds it).
#ifndef MOBI_DPARSE_TABLES_H
#define MOBI_DPARSE_TABLES_H
#include <stdint.h>

// byte offsets inside the table blob the host builds once (mobi_dparse_build_tables) and every workgroup copies to LDS
enum {
  MOBI_DT_A0 = 0,          // Vx2Table0_A  uint16[4096]   (MobiConst.cs)
  MOBI_DT_A1 = 8192,       // Vx2Table1_A
  MOBI_DT_B0 = 16384,      // Vx2Table0_B  uint8[256]
  MOBI_DT_B1 = 16640,      // Vx2Table1_B
  MOBI_DT_PLUT = 16896,    // partition-code lookup of this codec version, uint8[16][64]
  MOBI_DT_PBITS = 17920,   // code lengths uint8[16][12]
  MOBI_DT_PSHIFT = 18112,  // uint8[16]
  MOBI_DT_PNB = 18128,     // uint8[16]
  MOBI_DT_CBP_I = 18144,   // uint8[64]
  MOBI_DT_CBP_P = 18208,   // uint8[64]
  MOBI_DT_CBP4_I = 18272,  // uint8[20] (+12 pad)
  MOBI_DT_CBP4_P = 18304,  // uint8[16]
  MOBI_DT_ZZ8 = 18320,     // uint8[64]
  MOBI_DT_ZZ4 = 18384,     // uint8[16]
  MOBI_DT_BYTES = 18400
};
#define MOBI_LS_MAGIC 0x4C53u /* MobiDevResult.pad of a clip the lock-step parser (mobi_lsparse.hip) finished itself */
void mobi_dparse_build_tables(int version, uint8_t out[MOBI_DT_BYTES]); // mobi_parse.cpp (owns the tables)

#endif
