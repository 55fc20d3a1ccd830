/*
 * d2r_hip_probes.h - measurement aids of libd2r_hip.so (tests/probes/, bench.py's roofline leg).  NOT part of the drop-in surface
 * declared in d2r_hip.h: process-global state, not thread-safe, meant for one-process A/B runs and instrumented builds.
 */
#ifndef D2R_HIP_PROBES_H
#define D2R_HIP_PROBES_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Kernel-choice overrides for A/B measurements: LDS buffers of the generic kernel (1|2; bit 8: XCD-aware tile order off), vectorised
 * 16-bit epilogue (0|1), and a tile code: -1 automatic; 0..3 generic tiles 32x64 / 64x64 / 128x64 / 128x128; 4..9 LDS-DMA variants
 * (128x128 w4, 128x64, 128x128 w8, and their software-pipelined forms); 11 the 256 x 256 deep-pipelined kernel wherever eligible;
 * 100 / 101 128-wide grouped weight gradients off / on; 102 / 103 256-wide grouped weight gradients off / on; 110 / 111 automatic
 * 256-wide forward / dX products off / on; 120 / 121 grouped launches of d2r_gemm_group off / on; 1000 + n: fewest 256-wide tiles of
 * the automatic rule; 2000 + m: ablation mode of a -DD2R_GEMM_PROBES=1 build.  Defaults are the measured winners. */
void d2r_gemm_tuning(int nbuf, int vepi, int tile);
/* Measurement aid (bench.py's roofline leg; NOT part of the drop-in surface, not thread-safe against concurrent reads):
 * while on, d2r_gemm and d2r_gemm_tn_grouped - including the calls made inside the whole-layer / whole-module entry points -
 * bracket each launch with HIP events on the launching stream.  d2r_gemm_timer(1) clears and arms, d2r_gemm_timer(0)
 * disarms; d2r_gemm_timer_read waits for the recorded events and returns per launch: family = dtype * 8 + layout * 2 +
 * grouped + 100 * kernel variant (0 generic tiles, 1 LDS-DMA 128x64, 2 / 3 LDS-DMA 128x128 on four / eight waves, 20 grouped
 * LDS-DMA weight gradients, 21 grouped generic, 22 grouped batched 16-bit, 30 skinny fp32, 31 skinny 16-bit), flops, algorithmic bytes (operands
 * once, output once, twice when accumulated), milliseconds.  The single-head attention entry points record as well: family
 * 10001 = d2r_xattn_fwd_multi (one launch), 10002 = d2r_xattn_bwd_multi (its launches together; bytes = 2 x forward).  Returns the
 * number of records copied (or, with family == NULL, the number pending, which it discards). */
int d2r_gemm_timer(int on);
int d2r_gemm_timer_read(int* family, double* flops, double* bytes, float* ms, int capacity);


/* cycle stamps of workgroup 0 (measurement builds: D2R_GEMM_PROBES / D2R_G8_STAMPS / D2R_X3_PROBES=1 python -m d2r_amd.build) */
void d2r_gemm_debug_stamps(unsigned long long* dst);    /* LDS-DMA 128-wide kernel: [waves][8] */
void d2r_gemm8_debug_stamps(unsigned long long* dst);   /* 256-wide kernel: [8 waves][64] */
void d2r_xattn3_debug_stamps(unsigned long long* dst);  /* cross-attention forward: [4 waves][64] */
void d2r_xattn3_debug_mode(int mode);                   /* ablation mode of the cross-attention forward (measurement build) */

/* A/B of the AdamW kernel: non-temporal loads / stores of the fp32 streams (0 / 1), cap of the grid in workgroups (0 = default 2048). */
void d2r_adamw_probe_mode(int nt, int blocks);

#ifdef __cplusplus
}
#endif
#endif
