// Shared by the C-ABI translation units (api.hip: inference entry points, api_train.hip: training entry points).
#pragma once
#include "../../include/rtfs_amd.h"
#include "common.h"
#include "kernels.h"
#include <stdlib.h>

namespace {

constexpr int NF = 129;   // STFT bins (n_fft 256)
constexpr int CA = 256;   // audio feature channels
constexpr int CH = 64;    // block hidden channels
constexpr int FQ = 64;    // compressed frequency bins (n_freqs)

// ---------------------------------------------------------------- workspace carving
struct Arena {
    char* base;
    size_t off = 0, cap;
    Arena(void* b, size_t c) : base((char*)b), cap(c) {}
    template <class T>
    T* take(size_t n) {
        off = align_up(off, 256);
        T* r = base ? (T*)(base + off) : nullptr;
        off += n * sizeof(T);
        return r;
    }
    bool ok() const { return off <= cap; }
};

#define CHECK(expr)                  \
    do {                             \
        int _e = (expr);             \
        if (_e != RTFS_OK) return _e; \
    } while (0)

inline hipStream_t S(void* s) { return (hipStream_t)s; }

}  // namespace
