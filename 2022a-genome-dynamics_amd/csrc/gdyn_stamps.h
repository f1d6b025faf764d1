// gdyn_stamps.h -- in-kernel section stamps of the developer timing builds (make -C csrc abl N=30 / N=34; read back by
// gd_debug_bench, gdyn_dev.h).  In the product build every macro is empty: the kernels carry only the stamp points.
//   N = 30: k_step -- shader-clock cycles per section, one 16-word record per wave in the (unused in step mode) force buffer
//   N = 34: k_fill -- the same into BuildParams::dbg, plus two event counters
#pragma once
#ifndef GD_ABL
#define GD_ABL 0
#endif

#if GD_ABL == 30 || GD_ABL == 43      // (43: the ALU replay of gdyn_kernels.hip with the section stamps: where the arithmetic's issue time goes)
#define GD_STAMP_BEGIN() unsigned long long tprev_ = __builtin_amdgcn_s_memtime(), acc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define GD_STAMP(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc_[k] = now_ - tprev_; tprev_ = now_; } while (0)
#define GD_STAMP_USE3(a, b, c) asm volatile("" :: "v"(a), "v"(b), "v"(c))
#define GD_STAMP_END(buf) do { if ((threadIdx.x & 63) == 0) {                                                              \
        unsigned long long *rec_ = (buf) + ((size_t)blockIdx.x * (GD_BLOCK / 64) + (threadIdx.x >> 6)) * 16;              \
        for (int k_ = 0; k_ < 12; k_++) rec_[k_] = acc_[k_];                                                              \
        rec_[15] = 1ull; } } while (0)
#else
#define GD_STAMP_BEGIN() do { } while (0)
#define GD_STAMP(k) do { } while (0)
#define GD_STAMP_USE3(a, b, c) do { } while (0)
#define GD_STAMP_END(buf) do { } while (0)
#endif

#if GD_ABL == 34
#define GD_FSTAMP_BEGIN() unsigned long long ftprev_ = __builtin_amdgcn_s_memtime(), facc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define GD_FSTAMP(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); facc_[k] += now_ - ftprev_; ftprev_ = __builtin_amdgcn_s_memtime(); } while (0)
#define GD_FCOUNT(k) do { facc_[k] += 1; } while (0)
#define GD_FSTAMP_END(buf) do { if ((threadIdx.x & 63) == 0) {                                                             \
        unsigned long long *rec_ = (buf) + ((size_t)blockIdx.x * (GD_BLOCK / 64) + (threadIdx.x >> 6)) * 16;              \
        for (int k_ = 0; k_ < 12; k_++) rec_[k_] = facc_[k_];                                                             \
        rec_[15] = 1ull; } } while (0)
#else
#define GD_FSTAMP_BEGIN() do { } while (0)
#define GD_FSTAMP(k) do { } while (0)
#define GD_FCOUNT(k) do { } while (0)
#define GD_FSTAMP_END(buf) do { } while (0)
#endif
