// sw_scoring.h -- the align.Scoring handle shared by the SW score pass
// (sw_batch.hip) and the traceback (sw_traceback.hip).
#pragma once

#include <cstdint>

struct polyhip_scoring {
    int64_t gap;
    int32_t lut[65536];
    uint8_t validA[256], validB[256];
    int ncodes;          // valid A symbols
    uint8_t codeA[256];  // byte -> code, 0xFF = not in FirstAlphabet
    int ncodesB;         // valid B symbols
    uint8_t codeB[256];  // byte -> code, 0xFF = not in SecondAlphabet
    int cp;              // profile bytes per column (>= ncodes + 1, multiple of 4)
    int32_t smin, smax;  // over valid (a, b) pairs
    int32_t absmax;      // max(|smin|, |smax|, |gap|)
    bool int8_ok;
    int device;
    // device tables
    int8_t *d_lutc;      // [ncodes][256] int8 (only if int8_ok)
    uint8_t *d_codeA;    // [256]
    uint8_t *d_codeB;    // [256]
    int32_t *d_lutcc;    // [ncodes + 1][ncodesB + 1] compact int32 table (last row/col: zeros for pad codes)
    int32_t *d_lut;      // [256][256]
    uint8_t *d_validA, *d_validB;
};
