/*
 * fusion_hip_diag.h -- diagnostics of libfusion_hip.so: timers, per-dispatch profiling, launch-floor probes,
 * device-side launch timestamps, and reports of which HIP runtime / RCCL / Keccak variant serves the process.
 *
 * Nothing here replaces a reference interface (the reference's only instrumentation is the wall-clock wrapper of
 * benchmarks/benchmarks.py:25-34); these entries exist for bench.py, tools/ and the tests.  The surface a reference
 * maintainer binds is include/fusion_hip.h.  Same conventions as there: plain C, int status codes, fz_last_error().
 */
#ifndef FUSION_HIP_DIAG_H
#define FUSION_HIP_DIAG_H

#include "fusion_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* HIP version the library was built with and the one of the runtime it is bound to (e.g. 70200000 / 70051831), plus the
 * device's gcnArchName: which libamdhip64 a process ended up with is not always the one it was linked against. */
FZ_API int fz_runtime_info(fz_ctx *ctx, int *out_build_hip_version, int *out_runtime_hip_version, char *out_arch, size_t arch_cap);


/* which Keccak-f[1600] the host sponges run: "scalar", "bmi2" (the C form), "x64" (assembly block loop, BMI) or "x64v" (the same
 * with two of five rows per round in xmm registers, AVX-512VL) -- the fastest this CPU supports, measured once when
 * the library is loaded (FZ_KECCAK=<name> forces one); every variant is checked against the scalar one before it can be chosen */
FZ_API const char *fz_keccak_variant(void);

/* which RCCL serves fz_comm_* in this process (binds it if nothing has yet): the file the symbols came from, the rule that
 * found it -- "already mapped (shared)" (a copy some other component of the process loaded under the soname librccl.so.1,
 * e.g. torch's), "beside the HIP runtime" (the copy that ships next to the libamdhip64 the process runs on), "default search
 * path" -- and how many DIFFERENT files named librccl* the process has mapped (more than 1 = two copies of RCCL in one
 * process, the state the binding rule exists to avoid).  Any out pointer may be NULL. */
FZ_API int fz_rccl_library(char *out_path, size_t path_cap, char *out_how, size_t how_cap, int *out_copies_mapped);

/* ---- timing on the context's stream (hipEvent based) ----------------------------------- */
FZ_API int fz_timer_start(fz_ctx *ctx);
FZ_API int fz_timer_stop_ms(fz_ctx *ctx, float *out_ms);        /* records, waits, returns elapsed ms */

/* per-dispatch timing of the transform kernels: while enabled, every `sample_every`-th
 * fz_ntt_forward / fz_ntt_inverse / fz_ntt_multi launch carries a start/stop event pair bound to the dispatch
 * (kernel begin -> kernel end on the context's stream; at most max_launches pairs).
 * fz_profile_end synchronises and returns the average durations in microseconds. */
FZ_API int fz_profile_begin(fz_ctx *ctx, int max_launches, int sample_every);
FZ_API int fz_profile_end(fz_ctx *ctx, double *fwd_avg_us, int *fwd_count, double *inv_avg_us, int *inv_count);
/* the same, every sample: us[k] the duration of the k-th instrumented launch, kind[k] 0 = forward, 1 = inverse,
 * 2 = a multi-job launch (fz_ntt_multi) */
FZ_API int fz_profile_end_samples(fz_ctx *ctx, double *us, int *kind, int cap, int *n);

/* ---- launch-floor diagnostics (benchmarks) ---------------------------------------------------------------------
 * An empty 4096-workgroup dispatch and a plain 16-byte-per-lane copy on the context's stream: the two floors a
 * small-batch transform launch is judged against (bench.py reports them from the same run as the transforms). */
FZ_API int fz_diag_empty_launch(fz_ctx *ctx);
FZ_API int fz_diag_copy(fz_ctx *ctx, const void *d_src, void *d_dst, size_t bytes);
/* The shader clock the chip actually holds while the work already queued on the context's stream executes: one wave on a
 * private stream compares the shader cycle counter with the 100 MHz reference counter for `microseconds`, then the call
 * returns (synchronous).  The fp64-dense fused kernels run power-limited well below the nominal 2.4 GHz; a vector-issue
 * roofline has to be priced at THIS clock (profiles/README.md, round 3). */
FZ_API int fz_diag_shader_clock(fz_ctx *ctx, unsigned microseconds, double *out_mhz);
/* which transform schedule a launch of `rows` rows in all (one job, or the jobs of one fz_ntt_multi call together) takes on this
 * context: *family = 4 (radix-4 wave-tasks: ntt_fwd4 / ntt_inv4 / ntt_jobs4), 16 (16 coefficients per lane: ntt_fwd16 / ntt_inv16 /
 * ntt_jobs16) or 0 (another kernel: degrees outside 32..256).  What bench.py names its dominant kernel by, instead of mirroring
 * the library's crossover. */
FZ_API int fz_diag_ntt_schedule(fz_ctx *ctx, size_t rows, int *family);
/* one wave that occupies the context's stream for `microseconds` (asynchronous, capturable): a stand-in of known duration for
 * a step that cannot be run here -- bench.py uses it in place of the multi-GPU all-reduce to measure, on ONE GPU, how much of
 * an exchange step's latency its second stream hides */
FZ_API int fz_diag_delay(fz_ctx *ctx, unsigned microseconds);


/* ---- device-side launch timestamps of fz_ntt_multi ------------------------------------------------------------------
 * A profiler's kernel trace serialises the dispatches of all streams and HIP events are host-side markers: neither shows
 * when launches on DIFFERENT streams ran relative to each other.  While stamps are on, every workgroup of every
 * fz_ntt_multi launch of the context (degree 64 / 256) stores the chip-wide 100 MHz reference counter (s_memrealtime) at
 * entry and, after its stores have been acknowledged, at exit; launch k of the recording is [min entry, max exit] over its
 * workgroups.  Slots (16 bytes per workgroup) are assigned when a launch is issued or CAPTURED (the slot is part of the
 * recorded kernel arguments), up to max_launches launches / max_workgroups workgroups; later launches carry none.
 *   begin -> issue or capture the launches -> stop -> [reset -> ONE replay of the captured graph ->] read.
 * fz_diag_stamps_read synchronises the context's stream; h_start / h_end in ticks of 10 ns (0 / 0: no workgroup of the
 * launch has run since the reset); h_last_start (optional): the latest entry (when the launch's last workgroup was
 * dispatched); h_workgroups (optional): workgroups that stamped.  Stamped launches cost two scalar clock reads and one
 * 16-byte store per workgroup (bench.py never times them: the timed region runs without stamps). */
FZ_API int fz_diag_stamps_begin(fz_ctx *ctx, size_t max_launches, size_t max_workgroups);
FZ_API int fz_diag_stamps_stop(fz_ctx *ctx);
FZ_API int fz_diag_stamps_reset(fz_ctx *ctx);
FZ_API int fz_diag_stamps_read(fz_ctx *ctx, uint64_t *h_start, uint64_t *h_end, uint64_t *h_last_start, uint32_t *h_workgroups,
                               size_t cap, size_t *n);

#ifdef __cplusplus
}
#endif
#endif /* FUSION_HIP_DIAG_H */
