/*
 * pmhip.h -- C ABI of libpmhip.so: MI355X (gfx950) kernels for batched forward kinematics,
 * root-centred dual quaternions and the quaternion / dual-quaternion / 6D element-wise
 * conversions.  This is the drop-in boundary for the hot path of UPC-ViRVIG/pymotion
 * (pure Python, no FFI of its own): each entry point names the reference function
 * (file:line under pymotion/) whose semantics it reproduces.
 *
 * Conventions
 *   - All data pointers are DEVICE pointers to C-contiguous fp32 arrays unless the
 *     parameter is documented as "host".  The caller allocates and owns every buffer;
 *     the library never retains a caller pointer past the call's stream work.
 *   - Shapes use the reference's layout: joints axis = -2, components axis = -1,
 *     quaternions [w,x,y,z], matrices row-major [row][col], dual quats [qr(4), qd(4)],
 *     ortho6d [3][2].  Leading dims are flattened by the caller: F frames, N elements.
 *   - `parents` is a HOST int32[J] array; it is validated (parents[i] < i for i >= 1, the
 *     order the reference's in-place loops rely on, ops/skeleton.py:51-58) and passed to
 *     the kernel by value.  parents[0] is ignored like the reference does (:53).
 *   - Calls are asynchronous on `stream` (a hipStream_t; NULL = the default stream),
 *     re-entrant, and keep no global mutable state besides the thread-local error string.
 *   - Return 0 (PM_OK) or a negative PM_E* code; nothing throws or aborts across the ABI.
 *   - Fast path needs 16-byte aligned base pointers (anything from hipMalloc / torch is);
 *     other alignments take a scalar-access path with identical results.
 */
#ifndef PMHIP_H
#define PMHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PM_OK 0
#define PM_EINVAL (-1)       /* null pointer, negative size, J out of range            */
#define PM_ETOPOLOGY (-2)    /* parents not in topological order (parents[i] >= i)     */
#define PM_EHIP (-3)         /* a HIP runtime call failed (see pm_last_error_string)   */
#define PM_EUNSUPPORTED (-4) /* configuration the kernels do not cover                 */

#define PM_MAX_JOINTS 512

typedef void *pm_stream_t; /* hipStream_t */

/* ---- library / device plumbing ------------------------------------------------------------ */
int pm_version(void);                      /* ABI version, currently 1 */
const char *pm_last_error_string(void);    /* thread-local, valid until the next failing call */
const char *pm_last_kernel_name(void);     /* thread-local: the kernel the last skeleton-op call dispatched to,
                                              spelled as rocprofv3 prints it (bench.py's roofline.kernel) */
int pm_device_count(void);                 /* number of visible HIP devices, <0 on error */
int pm_set_device(int device);
int pm_get_device(int *device);
/* Plain device memory for callers without their own allocator (the NumPy front-end). */
int pm_malloc(void **dptr, size_t bytes);
int pm_free(void *dptr);
int pm_memcpy_h2d(void *dst, const void *src, size_t bytes, pm_stream_t stream);
int pm_memcpy_d2h(void *dst, const void *src, size_t bytes, pm_stream_t stream);
int pm_memset(void *dst, int value, size_t bytes, pm_stream_t stream);
int pm_stream_synchronize(pm_stream_t stream);
/* Time `fn`-agnostic sections on a stream with HIP events (used by bench.py). */
int pm_event_create(void **ev);
int pm_event_destroy(void *ev);
int pm_event_record(void *ev, pm_stream_t stream);
int pm_event_elapsed_ms(void *start, void *stop, float *ms); /* synchronises on `stop` */
int pm_event_synchronize(void *ev);
/* Streams and page-locked host memory for callers that overlap H2D / kernel / D2H themselves (the NumPy front-end's
 * chunked pipeline: pymotion_amd/_backend.py). */
int pm_stream_create(pm_stream_t *stream);
int pm_stream_destroy(pm_stream_t stream);
int pm_host_alloc(void **hptr, size_t bytes); /* hipHostMalloc: page-locked, DMA-able at full PCIe rate */
int pm_host_free(void *hptr);

/* ---- skeleton ops ---------------------------------------------------------------------------- */

/* ops/skeleton.py:16-61 / ops/skeleton_torch.py:16-66  fk(rot, global_pos, offsets, parents).
 * rot [F,J,4] (normalised internally, q/(|q|+1e-8)), root_pos [F,3], offsets [J,3] or, with
 * offsets_per_frame != 0, [F,J,3]; out pos [F,J,3], rotmats [F,J,3,3].  offsets[0] ignored. */
int pm_fk_f32(const float *rot, const float *root_pos, const float *offsets, int offsets_per_frame,
              const int32_t *parents /*host*/, int64_t F, int32_t J, float *pos, float *rotmats,
              pm_stream_t stream);

/* rotations/ortho6d.py:50-64 (to_quat) fused into fk (SURVEY config 4).  o6d [F,J,3,2];
 * eps: Gram-Schmidt denominators are max(norm, eps) -- 0 = NumPy reference behaviour,
 * 1e-12 = torch twin (ortho6d_torch.py:84-89).  quat_out [F,J,4] may be NULL. */
int pm_fk_from_ortho6d_f32(const float *o6d, const float *root_pos, const float *offsets,
                           int offsets_per_frame, const int32_t *parents /*host*/, int64_t F,
                           int32_t J, float eps, float *pos, float *rotmats, float *quat_out,
                           pm_stream_t stream);

/* ops/skeleton.py:207-244 / skeleton_torch.py:217-259  to_root_dual_quat(rotations, global_pos,
 * parents, offsets) -- note the argument order differs from fk.  rot [F,J,4] NOT normalised,
 * offsets [J,3] with offsets[0] == 0 (the caller checks, like the reference's assert :227);
 * out dq [F,J,8].  Joints whose parent is 0 stay local (:236-237). */
int pm_to_root_dq_f32(const float *rot, const float *root_pos, const int32_t *parents /*host*/,
                      const float *offsets, int64_t F, int32_t J, float *dq, pm_stream_t stream);

/* The same with a HOST-side hint: offsets_abs_max = max |offsets[j][k]| when the caller has the table on the host (the NumPy door
 * always has; the torch door remembers it per tensor), < 0 or NaN when unknown (= pm_to_root_dq_f32).  The library never reads
 * device memory to choose a kernel, so without the hint every skeleton decides per TILE (bones >= 1 unit or a root >= 16
 * units: float64 quaternion chain + fixed-point translations) on the kernels that are fastest on metre-scale data; with it, big-bone
 * skeletons (centimetre-scale BVH data) of 20 joints or more take the lane-per-frame kernel where the call has its 2.4 M joint-frames,
 * whose float64 state costs the same at every magnitude -- 5-25 % faster there than the per-tile precise step, most on skeletons of
 * twelve levels or more: a C caller who knows its bones are big should say so.  Results are within the same bar either way
 * (DESIGN.md 3a). */
int pm_to_root_dq_hint_f32(const float *rot, const float *root_pos, const int32_t *parents, const float *offsets, int64_t F,
                           int32_t J, float *dq, float offsets_abs_max, pm_stream_t stream);

/* ops/skeleton.py:173-204 / skeleton_torch.py:183-214  from_root_dual_quat(dq, parents)
 * -> (translations [F,J,3], rotations [F,J,4]) in that order (:204). */
int pm_from_root_dq_f32(const float *dq, const int32_t *parents /*host*/, int64_t F, int32_t J,
                        float *trans, float *rot, pm_stream_t stream);

/* ops/skeleton.py:64-93  from_global_rotations(global_quats, parents) -> local quats [F,J,4]. */
int pm_from_global_rotations_f32(const float *global_quats, const int32_t *parents /*host*/,
                                 int64_t F, int32_t J, float *local_quats, pm_stream_t stream);

/* ops/skeleton.py:247-344 mirror (modes 'all', 'symmetry') / :347-418 _true_mirror -- the rotation part,
 * fused: the result of fk -> quat.from_matrix -> gather joints_mapping -> negate two quaternion components ->
 * from_global_rotations from one kernel (world rotations are composed as quaternions and given the sign
 * from_matrix would pick; no matrix is formed).  mapping is a HOST int32[J] (NULL = identity, mode 'all');
 * axis 0/1/2 = X/Y/Z.  Translations / offsets / end sites are sign flips the front-end does. */
int pm_mirror_rotations_f32(const float *rot, const int32_t *parents /*host*/, const int32_t *mapping /*host*/,
                            int axis, int64_t F, int32_t J, float *out, pm_stream_t stream);

/* ops/skeleton.py:96-170 / skeleton_torch.py:101-176  from_root_positions(positions [F,J,3] root-centred,
 * parents, offsets [J,3]) -> local rotations [F,J,4]: align each joint's first child direction (from_to),
 * then correct the roll with every further child (from_to_axis).  One O(J) walk per frame instead of the
 * reference's fk-per-joint loop. */
int pm_from_root_positions_f32(const float *positions, const int32_t *parents /*host*/, const float *offsets,
                               int64_t F, int32_t J, float *rotations, pm_stream_t stream);

/* ---- element-wise conversions: N elements, inputs already broadcast by the caller ------------ */

/* rotations/quat.py:411-423  normalize(q, eps) = q / (|q| + eps) */
int pm_quat_normalize_f32(const float *q, int64_t N, float eps, float *out, pm_stream_t stream);
/* rotations/quat.py:364-376  length(q) -> [N] */
int pm_quat_length_f32(const float *q, int64_t N, float *out, pm_stream_t stream);
/* rotations/quat.py:276-317  to_matrix(q) -> [N,3,3] (no normalisation) */
int pm_quat_to_matrix_f32(const float *q, int64_t N, float *out, pm_stream_t stream);
/* rotations/quat.py:85-156  from_matrix(m [N,3,3]) -> [N,4] (4-branch select, then normalize) */
int pm_quat_from_matrix_f32(const float *m, int64_t N, float *out, pm_stream_t stream);
/* rotations/quat.py:337-361  mul(q0, q1) */
int pm_quat_mul_f32(const float *q0, const float *q1, int64_t N, float *out, pm_stream_t stream);
/* rotations/quat.py:320-334  mul_vec(q, v [N,3]) -> [N,3] */
int pm_quat_mul_vec_f32(const float *q, const float *v, int64_t N, float *out, pm_stream_t stream);
/* rotations/quat.py:396-408 (conjugate) and :379-393 (inverse == conjugate) */
int pm_quat_conjugate_f32(const float *q, int64_t N, float *out, pm_stream_t stream);

/* rotations/dual_quat.py:12-36  from_rotation_translation(q [N,4], t [N,3]) -> [N,8] */
int pm_dq_from_rt_f32(const float *q, const float *t, int64_t N, float *out, pm_stream_t stream);
/* rotations/dual_quat.py:62-83  to_rotation_translation(dq [N,8]) -> (q [N,4], t [N,3]) */
int pm_dq_to_rt_f32(const float *dq, int64_t N, float *q, float *t, pm_stream_t stream);
/* rotations/dual_quat.py:39-59  from_translation(t [N,3]) -> [N,8] */
int pm_dq_from_t_f32(const float *t, int64_t N, float *out, pm_stream_t stream);

/* rotations/ortho6d.py:67-90  to_matrix(x [N,3,2]) -> [N,3,3]; eps as in pm_fk_from_ortho6d_f32 */
int pm_o6d_to_matrix_f32(const float *x, int64_t N, float eps, float *out, pm_stream_t stream);
/* rotations/ortho6d.py:50-64  to_quat(x [N,3,2]) -> [N,4] */
int pm_o6d_to_quat_f32(const float *x, int64_t N, float eps, float *out, pm_stream_t stream);
/* rotations/ortho6d.py:14-28  from_quat(q) -> [N,3,2] */
int pm_o6d_from_quat_f32(const float *q, int64_t N, float *out, pm_stream_t stream);
/* rotations/ortho6d.py:31-47  from_matrix(m [N,3,3]) -> [N,3,2] (contiguous copy of m[..., :2]) */
int pm_o6d_from_matrix_f32(const float *m, int64_t N, float *out, pm_stream_t stream);

/* ---- second wave: trig conversions --------------------------------------------------------------- */

/* rotations/quat.py:24-40  from_angle_axis(angle [N,1], axis [N,3]) */
int pm_quat_from_angle_axis_f32(const float *angle, const float *axis, int64_t N, float *out, pm_stream_t stream);
/* rotations/quat.py:6-21  from_scaled_angle_axis(v [N,3]) (zero vector -> NaN, as the reference) */
int pm_quat_from_scaled_angle_axis_f32(const float *v, int64_t N, float *out, pm_stream_t stream);
/* rotations/quat.py:247-273  to_angle_axis(q) -> (angle [N,1], axis [N,3]) */
int pm_quat_to_angle_axis_f32(const float *q, int64_t N, float *angle, float *axis, pm_stream_t stream);
/* rotations/quat.py:230-244  to_scaled_angle_axis(q) -> [N,3] */
int pm_quat_to_scaled_angle_axis_f32(const float *q, int64_t N, float *out, pm_stream_t stream);
/* rotations/quat.py:43-82  from_euler(euler [N,3], order): order = device uint8 codes 0/1/2 for 'x'/'y'/'z':
 *   order_per_element == 0   one uint8[3] triple for every element,
 *   order_per_element == 1   uint8[N,3], an order per element (what the reference's signature literally asks for),
 *   order_per_element == P>=2  uint8[P,3], element e uses row e % P -- a [F, J, 3] clip with one order per joint
 *                            (P = J), which is what a BVH file gives and what tiling the order array F times means. */
int pm_quat_from_euler_f32(const float *euler, const uint8_t *order, int order_per_element, int64_t N,
                           float *out, pm_stream_t stream);
/* rotations/quat.py:159-227  to_euler(q, order) -> [N,3] in [0, 2pi) */
int pm_quat_to_euler_f32(const float *q, const uint8_t *order, int order_per_element, int64_t N,
                         float *out, pm_stream_t stream);
/* rotations/quat.py:465-501  slerp(q0, q1, t [N,1], shortest) */
int pm_quat_slerp_f32(const float *q0, const float *q1, const float *t, int64_t N, int shortest,
                      float *out, pm_stream_t stream);

/* rotations/quat.py:504-576  from_to(v1 [N,3], v2 [N,3], normalize_input) -> [N,4]: parallel -> identity,
 * anti-parallel -> half turn about an axis orthogonal to v1 (np.isclose thresholds) */
int pm_quat_from_to_f32(const float *v1, const float *v2, int64_t N, int normalize_input, float *out, pm_stream_t stream);
/* rotations/quat.py:579-650  from_to_axis(v1, v2, rot_axis [N,3], normalize_input) -> [N,4] */
int pm_quat_from_to_axis_f32(const float *v1, const float *v2, const float *axis, int64_t N, int normalize_input,
                             float *out, pm_stream_t stream);

/* rotations/quat.py:426-462  unroll(quaternions, axis): q is [T, S, 4] with the unroll axis FIRST (the
 * front-end moves it there); frame i is negated when the running sign says so: along T per series, a scan over the
 * maps {keep, negate, reset to +} that sgn(dot(q_i, q_{i-1})) selects (reset: a dot product of exactly 0 or NaN, where the
 * reference's `d0 < d1` is false whatever came before) -- one look-back kernel for S <= 64, three kernels beyond.
 * `workspace` is a device buffer of pm_quat_unroll_workspace_bytes(T, S) bytes, 8-byte aligned, contents irrelevant on
 * entry and on return; a call leaves nothing behind that the next call needs.  q and out 16-byte aligned.  pm_dq_unroll_f32 is
 * rotations/dual_quat.py:139-167: dq [T,S,8], sign decided by the real part, applied to all 8 floats. */
int64_t pm_quat_unroll_workspace_bytes(int64_t T, int32_t S);
int pm_quat_unroll_f32(const float *q, int64_t T, int32_t S, float *out, void *workspace, pm_stream_t stream);
int pm_dq_unroll_f32(const float *dq, int64_t T, int32_t S, float *out, void *workspace, pm_stream_t stream);
/* The same for a batch of clips, q [B, T, S, 4]: the layout of unroll(q, axis) whenever the axes in front of `axis` are batch
 * axes -- B independent scans in one launch, nothing is transposed (S <= 64: one look-back kernel for the whole batch;
 * beyond: the clips one after the other).  Workspace: pm_quat_unroll_batched_workspace_bytes(B, T, S), same contract. */
int64_t pm_quat_unroll_batched_workspace_bytes(int64_t B, int64_t T, int32_t S);
int pm_quat_unroll_batched_f32(const float *q, int64_t B, int64_t T, int32_t S, float *out, void *workspace, pm_stream_t stream);
int pm_dq_unroll_batched_f32(const float *dq, int64_t B, int64_t T, int32_t S, float *out, void *workspace, pm_stream_t stream);

/* rotations/dual_quat.py:86-136  normalize / is_unit.  The reference picks ONE branch for the whole batch
 * from global `.all()` reductions; the kernels raise three DEVICE ints that the caller zeroes first (pm_memset)
 * and reads back:  flags[0] != 0 iff some |qr|^2 !~ 0, flags[1] != 0 iff some |qr|^2 !~ 1, flags[2] != 0 iff some
 * qr.qd !~ 0 within atol  (np.isclose rules, NaN never close; the values are lower bounds of the violation counts).
 * pm_dq_normalize_f32: orthogonalize == 0 -> out = dq / |qr| and flags describe THAT result (:102-106);
 *                      orthogonalize != 0 -> the branch of :107-113 (flags may be NULL).
 * pm_dq_unit_flags_f32: flags of the input itself (is_unit, :118-136). */
int pm_dq_normalize_f32(const float *dq, int64_t N, int orthogonalize, float atol, float *out, int32_t *flags,
                        pm_stream_t stream);
int pm_dq_unit_flags_f32(const float *dq, int64_t N, float atol, int32_t *flags, pm_stream_t stream);

/* io/bvh.py:352-359 BVH.get_data: rots = quat.normalize(quat.unroll(quat.from_euler(np.radians(rotations), order), axis = 0)) in ONE
 * launch: euler_deg [T, J, 3] (DEVICE, the file's angles in degrees, fp32), order [J, 3] (HOST, axis codes 0 / 1 / 2 = x / y / z
 * per joint: the reference tiles this table over the frames), out [T, J, 4] unit, sign-unrolled quaternions.  12 B read + 16 B
 * written per joint and frame (the three launches it replaces: 28 + 32 + 32).  J <= 64 (PM_EUNSUPPORTED beyond: run the three
 * ops); workspace as for pm_quat_unroll_f32 (pm_quat_unroll_workspace_bytes(T, J)). */
int pm_bvh_rotations_f32(const float *euler_deg, const uint8_t *order, int64_t T, int32_t J, float *out, void *workspace,
                         pm_stream_t stream);

/* ---- resampling along the time axis ------------------------------------------------------------------ */

/* ops/time.py:4-66 interpolate_positions (method "linear"), torch twin ops/time_torch.py.
 * positions viewed as [A, T, B] (A = product of the axes before the time axis, B = product of those after),
 * out [A, S, B]:  out[a, s, :] = (1 - weights[s]) * positions[a, idx[s], :] + weights[s] * positions[a, idx[s] + 1, :]
 * idx (int32, 0 <= idx <= T - 2) and weights (fp32) are DEVICE arrays of S entries: the caller derives them
 * from the two 1-D time arrays exactly as time.py:49-54 does (searchsorted, clamp, divide).  T >= 2. */
int pm_interpolate_linear_f32(const float *positions, const int32_t *idx, const float *weights, int64_t A,
                              int64_t T, int64_t S, int64_t B, float *out, pm_stream_t stream);

/* ---- measurement helper --------------------------------------------------------------------------- */

/* Streaming ceiling with fk's traffic shape: per frame read rd_floats and write wr_floats
 * (contiguous tiles, no arithmetic).  Used only by bench.py to state what fraction of an
 * achievable copy rate the fk kernel reaches; not part of the reference's surface. */
int pm_stream_ceiling_f32(const float *src, float *dst, int64_t F, int32_t rd_floats,
                          int32_t wr_floats, pm_stream_t stream);

/* LDS-free streaming probe: n4 dwordx4 read, ratio * n4 written, grid-stride with `blocks` 256-thread blocks.
 * ratio = 0: pure read (n4 dwordx4 read, nothing written); ratio = -1: pure write (n4 dwordx4 written, nothing read).
 * Tells what the memory system sustains for a read:write mix (bench / tuning only). */
int pm_stream_plain_f32(const float *src, float *dst, int64_t n4, int32_t ratio, int32_t blocks, pm_stream_t stream);

/* Store-pattern probe (tools/store_probe.py -> profiles/r04_store_patterns.txt): n4 dwordx4 written in per-wave chunks under
 * the knobs of cfg, 12 HOST ints: [0] burst KiB per wave and chunk (1, 2, 3, 4, 6, 8, 12, 16, 24, 32), [1] dwordx4 per lane READ in front of a
 * chunk's stores (src must then hold n4 * cfg[1] / cfg[0] dwordx4), [2] store cache policy (0 plain, 1 nt, 2 sc1, 3 sc0 sc1,
 * 4 sc0 sc1 nt, 5 sc0, 6 sc1 nt, 10 / 11 buffer_store plain / nt, -1 hipMemsetD32Async instead of a kernel), [3] chunk -> address
 * placement (0 linear, 1 one contiguous range per XCD, 2 XCD-interleaved runs of cfg[4] block-chunks), [5] threads per workgroup
 * (64 / 256), [6] grid (0 = one block-chunk per workgroup), [7] 0 one output array, 1 two (3/4 + 1/4 of every chunk, fk's
 * rotmats + pos), 2 the two arrays one after the other, [8] what is stored (0 a per-chunk pattern plus what was read, 1 one constant, 2 random
 * bits), [9] 1 = plain instead of nt loads, [10] 1 = one load in flight per wave instead of all of a chunk's loads up front,
 * [11] bytes of unused LDS per workgroup (bounds the resident workgroups per CU like a kernel's LDS tile).  Bench / tuning only. */
int pm_store_probe_f32(const float *src, float *dst, int64_t n4, const int32_t *cfg, pm_stream_t stream);

/* The latency floor of a one-pass scan (tools/unroll_probe.py): the scan's grid and chain of dependencies -- ticket, one dwordx4 load per thread,
 * publish a status word, wait for the predecessor's, one dwordx4 store per thread -- and none of its work.  ntiles workgroups of `threads` (64, 128,
 * 192, 256) threads, tile t on dwordx4 [t * stride4, t * stride4 + threads) of src / dst; ws: ntiles + 129 32-bit words zero-filled once; epoch = 0, 1,
 * 2, ... call by call on the same ws, ntiles and mode (the words are self-resetting).  mode 0: one ticket counter for the launch, 1: no ticket
 * (tile = workgroup index), 2: one counter per XCD.  Bench / tuning only. */
int pm_scan_floor_probe(const float *src, float *dst, void *ws, int64_t ntiles, int64_t stride4, int32_t threads, uint32_t epoch, int32_t mode,
                        pm_stream_t stream);

/* The one-pass scans of quat.unroll / dual_quat.unroll / the BVH ingest WITHOUT the reset launch in front of them (reference: rotations/quat.py:426-462,
 * dual_quat.py:139-167, io/bvh.py:352-359; a clip of real length is 7-19 us on the device of which that launch is ~3).  The caller owns a PAIR of
 * workspaces per stream -- each at least pm_quat_unroll_batched_workspace_bytes(B, T, S) bytes, 8-byte aligned, both zero-filled once (pm_memset) --
 * and alternates them: `ws_zeroed` is the clean one, `ws_other` the one the call before left dirty.  This launch zeroes the first `ws_other_words`
 * 8-byte words of `ws_other` on the way and reports in *ws_words_dirtied how many words of `ws_zeroed` it leaves non-zero -- the count to pass as
 * ws_other_words of the NEXT call, with the roles swapped.  kind: 0 quat.unroll, 1 dual_quat.unroll (in / out [B, T, S, 4 | 8], clips scanned
 * independently), 2 the BVH ingest (B = 1, in = Euler degrees [T, S, 3], `order` as for pm_bvh_rotations_f32; otherwise null).  S <= 64
 * (PM_EUNSUPPORTED beyond: pm_quat_unroll_f32's three-pass scan has no reset to save).  A workspace that is not zero on entry makes the scan
 * wait for words nobody writes: the pair is the caller's to keep private to one stream.  On PM_OK (an empty scan included) `ws_other` is clean and the
 * roles swap.  On PM_EINVAL / PM_EUNSUPPORTED (argument checks, before any launch) neither workspace was touched, though *ws_words_dirtied may have
 * been written; after PM_EHIP (a failed launch) the state of both is unknown: zero-fill them again, or drop the pair. */
int pm_unroll_onepass_f32(int32_t kind, const float *in, const uint8_t *order, int64_t B, int64_t T, int32_t S, float *out, void *ws_zeroed,
                          int64_t *ws_words_dirtied, void *ws_other, int64_t ws_other_words, pm_stream_t stream);

/* Host only (no GPU work): the step list fk's wide walk would run for this topology (fkwide.hip: a wave per frame, up to 16 joints a
 * step, a joint at the earliest one step after its parent) -- `jobs` receives (steps + 2) * 16 words, joint | parent << 16 (the root takes no step;
 * idle quads: J + 1 | J << 16), room for 50 * 16.  Returns the number of steps, PM_EUNSUPPORTED when the tree needs more
 * than the kernel's list holds (fk then keeps its tile kernels), another error code on a bad topology.  For tests of the scheduler; the reference has no
 * counterpart (ops/skeleton.py:51-58 is a loop over joints). */
int pm_fk_wide_plan_debug(const int32_t *parents, int32_t J, uint32_t *jobs);

/* Host only (no GPU work): the step words the step-list kernels of round 6 would run for this topology -- op 0: to_root_dual_quat (dqwide.hip; ops/skeleton.py:230-241),
 * op 1: mirror (mirror.hip; ops/skeleton.py:300-331) -- at `fpw` = 1 / 2 / 4 / 8 frames a wave, i.e. 16 / fpw joints of a frame a step.  `jobs` receives 16 x 56 words,
 * jobs[k * 56 + step] = own slot | parent slot << 16 in BYTES of the frame's image (32-byte slots for op 0, 16-byte slots for op 1): slot J is the identity (op 0: what
 * the root's children compose with -- they stay local, skeleton.py:236-237), slot J + 1 what idle quads read and write; the root takes no step.  Returns the number of
 * steps, PM_EUNSUPPORTED when the tree needs more than the 48 the list holds (the dispatch then keeps the other kernels), another error code on a bad topology or
 * argument.  For tests of the scheduler; the reference has no counterpart (a loop over joints). */
int pm_step_list_plan_debug(const int32_t *parents, int32_t J, int32_t op, int32_t fpw, uint32_t *jobs);

#ifdef __cplusplus
}
#endif
#endif /* PMHIP_H */
