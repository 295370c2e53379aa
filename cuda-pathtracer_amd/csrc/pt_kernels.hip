// pt_kernels.hip — the gfx950 per-pixel path-tracing megakernel and its launchers.
//
// Replaces the reference's `__global__ kernel` (cuda_opengl/src/shaders/raytrace.cu:212-271)
// together with radiance() (:41-210) and intersect() (intersection.cuh:161-246).
//
// Shape of the kernel (MI355X-first, not a translation of the 16x16-thread CUDA launch):
//   * one wave64 owns an 8x8 pixel tile; the default (persistent) kernel keeps 24 waves per CU
//     resident for the whole launch and hands them tiles from eight per-XCD ticket heads;
//   * at workgroup start the whole traversal working set — BVH nodes and 48-byte {e1,e2,v0}
//     triangle records — is staged into LDS, the nodes re-encoded on the way (32-byte boxes +
//     16-bit hit/miss link codes, see walk_to_leaf_lds), so the hot loop never touches HBM/L2
//     (scenes that do not fit fall back to L2-resident global reads of the plain 64-byte nodes);
//   * nearest hit = stackless, per-octant ORDERED threaded BVH walk ("while-while": all
//     lanes walk boxes until each has a leaf, then all lanes test triangles), conservative
//     boxes, (t, global face index) lexicographic minimum == the reference's first-wins
//     brute-force loop;  the brute-force loop itself is kept as a second variant with
//     wave-uniform LDS broadcast reads (zero traversal divergence);
//   * shading data (normals/uvs/material) is fetched from L2 only for the winning face;
//   * a ray that missed cannot hit later (the reference keeps looping with the same ray,
//     raytrace.cu:194-199), so its remaining iterations skip the walk — exact.
#include "pt_device.h"
#include "pt_launch.h"

namespace ptamd {

#ifndef PT_TILE_THREADS
#define PT_TILE_THREADS 512
#endif
#ifndef PT_TILE_WAVES_PER_EU
#define PT_TILE_WAVES_PER_EU 6
#endif
#ifndef PT_SP_THREADS
#define PT_SP_THREADS 768
#endif
#ifndef PT_SP_SHADERS
#define PT_SP_SHADERS 7
#endif
#ifndef PT_SP_WAVES_PER_EU
#define PT_SP_WAVES_PER_EU 6
#endif
#ifndef PT_SP_FETCH_MIN
#define PT_SP_FETCH_MIN 16u
#endif
#ifndef PT_BW_THREADS
#define PT_BW_THREADS 512
#endif
#ifndef PT_BW_WAVES_PER_EU
#define PT_BW_WAVES_PER_EU 4
#endif
#ifndef PT_BW_FETCH_MIN
#define PT_BW_FETCH_MIN 16u
#endif
#ifndef PT_PERSISTENT_WAVES_PER_EU
#define PT_PERSISTENT_WAVES_PER_EU 6
#endif
#ifndef PT_PERSISTENT_THREADS
#define PT_PERSISTENT_THREADS 512
#endif
#ifndef PT_RS_THREADS
#define PT_RS_THREADS 768   /* two workgroups of 12 waves per CU: two scene copies leave room for the waves' pools in LDS */
#endif
#ifndef PT_RS_WAVES_PER_EU
#define PT_RS_WAVES_PER_EU 6
#endif
#ifndef PT_CAM_FROM_KERNARG
#define PT_CAM_FROM_KERNARG 1
#endif
#ifndef PT_LEAF_MIN
#define PT_LEAF_MIN 64u /* lanes parked at a leaf that end a box phase early (64 = only when all are parked) */
#endif
#ifndef PT_ASM_WALK
#define PT_ASM_WALK 1
#endif
#ifndef PT_ASM_LEAF_WIDE
#define PT_ASM_LEAF_WIDE 0
#endif
#ifndef PT_ASM_LEAF_EXITS
#define PT_ASM_LEAF_EXITS 7
#endif
#ifndef PT_ASM_LEAF
#define PT_ASM_LEAF 1 /* hand-scheduled Moller-Trumbore (mt_test_asm) in the leaf phases of the LDS-resident restart kernel */
#endif
#ifndef PT_WALK_PRIO
#define PT_WALK_PRIO 3 /* s_setprio inside the hand-scheduled box loop: waves that walk win VALU arbitration (+0.7 %) */
#endif
#define PT_STR2(x) #x
#define PT_STR(x) PT_STR2(x)
#define PT_MAX_DIST 100000.0f
#define PT_END 0xFFFFFFFFu

// ---------------------------------------------------------------- nearest hit

struct Best { float t, u, v; uint32_t idx; };

// intersection.cuh:102-135 on a {e1,e2,v0} record; identical operation order.
// Accept rule: reference `t < best && t > 0` in storage order == lexicographic (t, idx).
// small_det (wave-uniform, from the launcher): every determinant of this scene stays below 2^125 (coordinates <= 1e8),
// so the survivors of the det >= 1e-7 test are inside rcp_exact_in_range's domain.
template <bool ORDERED>
PT_DEV void mt_test(float4 a, float4 b, float4 c, f3 o, f3 d, Best& best, bool small_det = false)
{
  const f3 e1 = mk3(a.x, a.y, a.z);
  const f3 e2 = mk3(a.w, b.x, b.y);
  const f3 v0 = mk3(b.z, b.w, c.x);
  const f3 p_vec = cross(d, e2);
  const float det = dot(e1, p_vec);
  // intersection.cuh:110 compares in double: (double)det < 1e-7.  1e-7f is the float nearest to
  // (and above) the double 1e-7 and no float lies between them, so `det < 1e-7f` decides identically.
  if (det < 1e-7f) return;
  const float inv_det = small_det ? rcp_exact_in_range(det) : 1.0f / det;
  const f3 t_vec = o - v0;
  const float u = dot(t_vec, p_vec) * inv_det;
  if (u < 0 || u > 1) return;
  const f3 qvec = cross(t_vec, e1);
  const float v = dot(d, qvec) * inv_det;
  if (v < 0 || u + v > 1) return;
  const float t = dot(e2, qvec) * inv_det;
  const uint32_t idx = f_as_u(c.y);
  bool take;
  if (ORDERED) take = t < best.t && t > 0.0f;
  else take = t > 0.0f && (t < best.t || (t == best.t && idx < best.idx && best.idx != PT_END));
  if (take) { best.t = t; best.u = u; best.v = v; best.idx = idx; }
}

// mt_test<false> for scenes with small determinants (rcp_exact_in_range), hand-scheduled: the same operations in the same order, but a
// lane that fails a test leaves by clearing its EXEC bit with the comparison itself (v_cmpx) instead of through hipcc's
// s_and_saveexec / s_cbranch_execz / s_or_b64 group per early exit and its s_and / s_or chains for the (t, index) rule: 2 scalar
// instructions per triangle instead of ~30.  The CU's single scalar pipe issues half as many instructions per clock as its vector
// pipes and the resident kernel keeps both ~58 % busy (scripts/ubench/ifetch_rate.hip, profiles/r04_notes.md).
// NaN behaviour is the reference's: every rejection is written as the negation it is in intersection.cuh:110-128
// (`det < eps`, `u < 0 || u > 1`, `v < 0 || u + v > 1` reject; an unordered comparison does not), `t > 0` accepts.
// The (t, index) rule `t < best.t || (t == best.t && idx < best.idx && best.idx != PT_END)`: among lanes with t <= best.t,
// idx + 1 < (t == best.t ? best.idx + 1 : 0xFFFFFFFF) as unsigned numbers (PT_END + 1 wraps to 0: no index is below it).
PT_DEV void mt_test_asm(float4 a, float4 b, float2 c, f3 o, f3 d, Best& best)
{
  float px, py, pz, qx, qy, qz, det, inv, u, v, t0, t1;
  float v0x = b.z, v0y = b.w, v0z = c.x;   // overwritten by t_vec
  unsigned long long save;
  asm volatile(
      "s_mov_b64 %[save], exec\n\t"
      // p_vec = cross(d, e2)
      "v_mul_f32 %[px], %[dy], %[e2z]\n\t"
      "v_mul_f32 %[t0], %[dz], %[e2y]\n\t"
      "v_mul_f32 %[py], %[dz], %[e2x]\n\t"
      "v_mul_f32 %[t1], %[dx], %[e2z]\n\t"
      "v_sub_f32 %[px], %[px], %[t0]\n\t"
      "v_sub_f32 %[py], %[py], %[t1]\n\t"
      "v_mul_f32 %[pz], %[dx], %[e2y]\n\t"
      "v_mul_f32 %[t0], %[dy], %[e2x]\n\t"
      "v_sub_f32 %[pz], %[pz], %[t0]\n\t"
      // det = dot(e1, p_vec)
      "v_mul_f32 %[t0], %[e1x], %[px]\n\t"
      "v_mul_f32 %[t1], %[e1y], %[py]\n\t"
      "v_add_f32 %[t0], %[t0], %[t1]\n\t"
      "v_mul_f32 %[t1], %[e1z], %[pz]\n\t"
      "v_add_f32 %[det], %[t1], %[t0]\n\t"
      "v_cmpx_ngt_f32 vcc, 0x33d6bf95, %[det]\n\t"       // keep !(1e-7f > det)
#if PT_ASM_LEAF_EXITS & 1
      "s_cbranch_execz 9f\n\t"
#endif
      // inv_det = rcp_exact_in_range(det); t_vec = o - v0 fills the wait state a transcendental's consumer needs
      "v_rcp_f32 %[inv], %[det]\n\t"
      "v_sub_f32 %[v0x], %[ox], %[v0x]\n\t"
      "v_sub_f32 %[v0y], %[oy], %[v0y]\n\t"
      "v_sub_f32 %[v0z], %[oz], %[v0z]\n\t"
      "v_fma_f32 %[t0], -%[det], %[inv], 1.0\n\t"
      "v_fma_f32 %[inv], %[t0], %[inv], %[inv]\n\t"
      "v_fma_f32 %[t0], -%[det], %[inv], 1.0\n\t"
      "v_fma_f32 %[t0], %[t0], %[inv], %[inv]\n\t"
      "v_fma_f32 %[t1], -%[det], %[t0], 1.0\n\t"
      "v_fma_f32 %[inv], %[t1], %[inv], %[t0]\n\t"
      // u = dot(t_vec, p_vec) * inv_det
      "v_mul_f32 %[t0], %[v0x], %[px]\n\t"
      "v_mul_f32 %[t1], %[py], %[v0y]\n\t"
      "v_add_f32 %[t0], %[t0], %[t1]\n\t"
      "v_mul_f32 %[t1], %[pz], %[v0z]\n\t"
      "v_add_f32 %[t0], %[t1], %[t0]\n\t"
      "v_mul_f32 %[u], %[t0], %[inv]\n\t"
      "v_cmpx_ngt_f32 vcc, 0, %[u]\n\t"                  // keep !(u < 0)
      "v_cmpx_nlt_f32 vcc, 1.0, %[u]\n\t"                // keep !(u > 1)
#if PT_ASM_LEAF_EXITS & 2
      "s_cbranch_execz 9f\n\t"
#endif
      // qvec = cross(t_vec, e1)
      "v_mul_f32 %[qx], %[e1z], %[v0y]\n\t"
      "v_mul_f32 %[t0], %[e1y], %[v0z]\n\t"
      "v_mul_f32 %[qy], %[e1x], %[v0z]\n\t"
      "v_mul_f32 %[t1], %[e1z], %[v0x]\n\t"
      "v_sub_f32 %[qx], %[qx], %[t0]\n\t"
      "v_sub_f32 %[qy], %[qy], %[t1]\n\t"
      "v_mul_f32 %[qz], %[e1y], %[v0x]\n\t"
      "v_mul_f32 %[t0], %[e1x], %[v0y]\n\t"
      "v_sub_f32 %[qz], %[qz], %[t0]\n\t"
      // v = dot(d, qvec) * inv_det
      "v_mul_f32 %[t0], %[dx], %[qx]\n\t"
      "v_mul_f32 %[t1], %[dy], %[qy]\n\t"
      "v_add_f32 %[t0], %[t0], %[t1]\n\t"
      "v_mul_f32 %[t1], %[dz], %[qz]\n\t"
      "v_add_f32 %[t0], %[t1], %[t0]\n\t"
      "v_mul_f32 %[v], %[t0], %[inv]\n\t"
      "v_cmpx_ngt_f32 vcc, 0, %[v]\n\t"                  // keep !(v < 0)
      "v_add_f32 %[t0], %[u], %[v]\n\t"
      "v_cmpx_nlt_f32 vcc, 1.0, %[t0]\n\t"               // keep !(u + v > 1)
#if PT_ASM_LEAF_EXITS & 4
      "s_cbranch_execz 9f\n\t"
#endif
      // t = dot(e2, qvec) * inv_det
      "v_mul_f32 %[t0], %[e2x], %[qx]\n\t"
      "v_mul_f32 %[t1], %[e2y], %[qy]\n\t"
      "v_add_f32 %[t0], %[t0], %[t1]\n\t"
      "v_mul_f32 %[t1], %[e2z], %[qz]\n\t"
      "v_add_f32 %[t0], %[t1], %[t0]\n\t"
      "v_mul_f32 %[det], %[t0], %[inv]\n\t"               // t (in det's register)
      "v_cmpx_lt_f32 vcc, 0, %[det]\n\t"                  // keep t > 0
      "v_cmpx_le_f32 vcc, %[det], %[bt]\n\t"              // keep t <= best.t
      "v_cmp_eq_f32 vcc, %[det], %[bt]\n\t"
      "v_add_u32 %[t0], 1, %[idx]\n\t"
      "v_add_u32 %[t1], 1, %[bi]\n\t"
      "v_cndmask_b32 %[t1], -1, %[t1], vcc\n\t"           // (two VALU since the compare: the wait states a VCC reader needs)
      "v_cmpx_lt_u32 vcc, %[t0], %[t1]\n\t"
      "v_mov_b32 %[bt], %[det]\n\t"
      "v_mov_b32 %[bu], %[u]\n\t"
      "v_mov_b32 %[bv], %[v]\n\t"
      "v_mov_b32 %[bi], %[idx]\n\t"
      "9:\n\t"
      "s_mov_b64 exec, %[save]\n\t"
      : [bt] "+v"(best.t), [bu] "+v"(best.u), [bv] "+v"(best.v), [bi] "+v"(best.idx), [v0x] "+v"(v0x), [v0y] "+v"(v0y), [v0z] "+v"(v0z),
        [px] "=&v"(px), [py] "=&v"(py), [pz] "=&v"(pz), [qx] "=&v"(qx), [qy] "=&v"(qy), [qz] "=&v"(qz), [det] "=&v"(det), [inv] "=&v"(inv),
        [u] "=&v"(u), [v] "=&v"(v), [t0] "=&v"(t0), [t1] "=&v"(t1), [save] "=&s"(save)
      : [e1x] "v"(a.x), [e1y] "v"(a.y), [e1z] "v"(a.z), [e2x] "v"(a.w), [e2y] "v"(b.x), [e2z] "v"(b.y), [idx] "v"(f_as_u(c.y)),
        [ox] "v"(o.x), [oy] "v"(o.y), [oz] "v"(o.z), [dx] "v"(d.x), [dy] "v"(d.y), [dz] "v"(d.z)
      : "vcc");
}

// The reference algorithm: every face in storage order.  `tris` is LDS (or global); the
// index is wave-uniform so LDS reads are broadcasts.
template <bool STATS>
PT_DEV void traverse_brute(const float4* tris, uint32_t n_faces, f3 o, f3 d, Best& best,
                           uint32_t& n_tris)
{
  for (uint32_t i = 0; i < n_faces; ++i) {
    const float4 a = tris[i * 3 + 0], b = tris[i * 3 + 1], c = tris[i * 3 + 2];
    mt_test<true>(a, b, c, o, d, best);
  }
  if (STATS) n_tris += n_faces;
}

// Stackless ordered threaded BVH walk (layout: host/ptamd_internal.h).
// The box test only has to be CONSERVATIVE (never cull the winner), not bit-reproducible, so it
// uses v_rcp_f32 and one fma per slab plane; boxes carry a 1 mm margin that dwarfs its rounding
// (DESIGN.md "Conservative boxes").  Everything that decides the RESULT — Moller-Trumbore and
// the (t, index) minimum — is the reference's exact operation sequence (mt_test).
struct Walk {
  f3 o, d, inv, noi; // ray, 1/d, -o/d
  uint32_t link_off; // dword offset of this ray's octant inside a node's miss-link table
  uint32_t n_nodes;  // compact LDS layout: the link array starts 32 * n_nodes bytes after the box array
  uint32_t oct;
  unsigned long long oct_x, oct_y, oct_z;   // wide walk: the lanes whose ray runs against x / y / z (ballots, set by traverse_round4)
  uint32_t node;     // next node to test, PT_END when the walk is over
  Best best;
  // STATS builds only: where the idle lane-slots of the box loop come from
  uint32_t alive;    // lanes of the wave that trace a ray in this call
  uint32_t idle_unstarted, idle_finished, idle_parked;
};

// Slab distances of the hand-scheduled LDS box loop are in units of 2^17 (an exact scaling): MAX_DIST = 1e5 then lies below 1, and
// the lower clamp of a box's entry distance at 0 is the clamp modifier of one of the loop's fma — one instruction less per iteration.
#define PT_T_SCALE 7.62939453125e-06f   /* 2^-17 */
// asm_scaled: the walk is going to run through walk_to_leaf_lds_state (which expects inv / noi in those units)
PT_DEV void walk_init(Walk& w, f3 o, f3 d, uint32_t n_nodes, bool asm_scaled = false)
{
  w.o = o; w.d = d;
  w.oct = (d.x < 0.f ? 1u : 0u) | (d.y < 0.f ? 2u : 0u) | (d.z < 0.f ? 4u : 0u);
  // a zero component would make lo*inv - o*inv an inf - inf; a denormal-sized stand-in keeps the
  // slab arithmetic finite and classifies "origin inside/outside the slab" the same way
  const float tiny = 1e-30f;
  const float dx = __builtin_fabsf(d.x) < tiny ? __builtin_copysignf(tiny, d.x) : d.x;
  const float dy = __builtin_fabsf(d.y) < tiny ? __builtin_copysignf(tiny, d.y) : d.y;
  const float dz = __builtin_fabsf(d.z) < tiny ? __builtin_copysignf(tiny, d.z) : d.z;
  const float ts = asm_scaled ? PT_T_SCALE : 1.0f;
  w.inv = mk3(__builtin_amdgcn_rcpf(dx) * ts, __builtin_amdgcn_rcpf(dy) * ts, __builtin_amdgcn_rcpf(dz) * ts);
  w.noi = mk3(-(o.x * w.inv.x), -(o.y * w.inv.y), -(o.z * w.inv.z));
  w.oct_x = w.oct_y = w.oct_z = 0ull;
  w.link_off = 8u + w.oct;
  w.n_nodes = n_nodes;
  w.node = n_nodes ? 0u : PT_END;
  w.best.t = PT_MAX_DIST; w.best.u = 0.f; w.best.v = 0.f; w.best.idx = PT_END;
  w.alive = 64u; w.idle_unstarted = w.idle_finished = w.idle_parked = 0u;
}

// Box tests until this lane holds a leaf (leaf_count != 0) or its walk is over.
// walk_min > 1: the box phase also ends once fewer than walk_min lanes of the wave are still walking (the rest are
// parked at a leaf or done); the walkers keep their place.
template <bool STATS>
PT_DEV void walk_to_leaf(const float4* nodes, Walk& w, uint32_t& leaf_first, uint32_t& leaf_count,
                         uint32_t& n_nodes_visited, uint32_t& wave_node_iters, uint32_t walk_min = 1u)
{
  const float* links = reinterpret_cast<const float*>(nodes) + w.link_off;
  leaf_first = 0;
  leaf_count = 0;
  uint32_t node = w.node;
  const uint32_t lanes_in = (uint32_t)__popcll(__ballot(node != PT_END)); // walkers entering the box phase
  while (node != PT_END) {
    const float4 q0 = nodes[node * 4 + 0];
    const float4 q1 = nodes[node * 4 + 1];
    const uint32_t miss = f_as_u(links[node * 16]);
    if (STATS) {
      ++n_nodes_visited;
      // one lane per executing wave counts the wave-level iteration (lane utilisation = nodes / (64 * iters))
      const unsigned long long act = __ballot(1);
      if ((threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)act) - 1)) {
        ++wave_node_iters;
        const uint32_t active = (uint32_t)__popcll(act);
        w.idle_unstarted += 64u - w.alive;       // no ray in this call (dead path, edge of the frame)
        w.idle_finished += w.alive - lanes_in;   // walk already over, waiting for the wave
        w.idle_parked += lanes_in - active;      // left this box phase (leaf found or walk just ended)
      }
    }
    const float t0x = __builtin_fmaf(q0.x, w.inv.x, w.noi.x), t1x = __builtin_fmaf(q1.x, w.inv.x, w.noi.x);
    const float t0y = __builtin_fmaf(q0.y, w.inv.y, w.noi.y), t1y = __builtin_fmaf(q1.y, w.inv.y, w.noi.y);
    const float t0z = __builtin_fmaf(q0.z, w.inv.z, w.noi.z), t1z = __builtin_fmaf(q1.z, w.inv.z, w.noi.z);
    const float tnear = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t0x, t1x), __builtin_fminf(t0y, t1y)),
                                        __builtin_fminf(t0z, t1z));
    const float tfar = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t0x, t1x), __builtin_fmaxf(t0y, t1y)),
                                       __builtin_fmaxf(t0z, t1z));
    const bool hit = tnear <= tfar && tfar >= 0.0f && tnear <= w.best.t;
    const uint32_t info = f_as_u(q0.w);
    const uint32_t child = f_as_u(q1.w);
    const uint32_t count = info >> 24;
    const uint32_t down = ((w.oct >> (child >> 30)) & 1u) ? (child & 0x3FFFFFFFu) : node + 1u;
    node = (hit && count == 0u) ? down : miss;
    if (hit && count != 0u) {
      leaf_first = info & 0xFFFFFFu;
      leaf_count = count;
      break;
    }
    // leave the box phase early once PT_LEAF_MIN lanes of the wave are parked at a leaf: they
    // would otherwise idle until the slowest walker finds its own (the walkers keep their place)
    if (PT_LEAF_MIN < 64 && (uint32_t)__popcll(__ballot(1)) + PT_LEAF_MIN <= lanes_in) break;
    if (walk_min > 1u && (uint32_t)__popcll(__ballot(1)) < walk_min) break;
  }
  w.node = node;
}

// The same loop for LDS-resident nodes, hand-scheduled.  hipcc's version of the C++ loop above
// carries ~20 SALU exec-mask instructions per iteration (two exits, phi merges of masks) next to
// ~35 VALU, and the rocprofv3 counters show 39 % of wave time stalled at issue.  This version
// keeps one exec update per iteration: lanes leave the loop by clearing their exec bit when they
// reach a leaf they hit or run off the tree.  v64..v69 and v74 are scratch (clobbered); masks live in
// compiler-allocated SGPR pairs.  Hazards: a VALU that reads an SGPR mask written by a VALU
// compare needs 2 wait states (s_nop 1), exactly as hipcc pads it; SALU consumers are interlocked.
// `state` is the walk's position in the loop's own terms (below): the LDS address of the next box, or 0xFFFF when the walk is
// over; it stays in that form between the box phases of a round (traverse_round) — converting it to a node index and
// back costs four VALU per phase.
// The loop names its scratch registers (v64..v69, v74: a ds_read_b128 needs four consecutive ones, which an inline-asm operand
// cannot express dword by dword); they are in the clobber list, so the allocator keeps everything else out of them.  They must exist
// in every kernel that inlines this loop: 75 VGPRs at least, i.e. at most 6 waves per SIMD (512 / 6 = 85) — checked here at build time.
static_assert(PT_TILE_WAVES_PER_EU <= 6 && PT_PERSISTENT_WAVES_PER_EU <= 6 && PT_RS_WAVES_PER_EU <= 6 && PT_SP_WAVES_PER_EU <= 6,
              "walk_to_leaf_lds_state clobbers v64..v74: kernels that inline it need a budget of at least 75 VGPRs (<= 6 waves per SIMD)");
PT_DEV void walk_to_leaf_lds_state(uint32_t& state, uint32_t lnk, const Walk& w, uint32_t& leaf_first, uint32_t& leaf_count, uint32_t walk_min)
{
  // COMPACT LDS nodes (stage_scene), two arrays of 32-byte records, N nodes each:
  //   boxes[n] at lds_nodes + 32 n:          dwords 0..3 centre.xyz half.x | 4..5 half.yz | 6 leaf word
  //   links[n] at lds_nodes + 32 N + 32 n:   one word per ray octant
  // links[n][o] = hit code | miss code << 16, 16 bits each:
  //   < 0x8000  LDS byte address of the box to test next          0xFFFF  end of the walk
  //   0x8000 | count << 11 | first   "park: test these triangle records" (only ever a hit code, of the leaf itself)
  // so one v_cndmask with sub-dword selects yields the next state and one unsigned compare says whether to keep
  // walking: 16 VALU, 3 SALU and 8 LDS cycles per iteration (the plain layout took 33 / 20 / 10; lo / hi boxes 20 VALU).  A lane that parks
  // stops executing: the link word it read last is its leaf's, whose upper half says where the walk goes on after the tests.
  unsigned long long save;
  uint32_t link_word; // the link word a lane read last (a lane that parks read its leaf's)
  uint32_t walkers;   // lanes still in the box loop; the loop runs while walkers >= walk_min (walk_min 1: until none is left)
  asm volatile(
      "s_mov_b64 %[save], exec\n\t"
#ifdef PT_WALK_PRIO
      "s_setprio " PT_STR(PT_WALK_PRIO) "\n\t"
#endif
      "v_cmp_gt_u32 vcc, 0x8000, %[st]\n\t"
      "s_and_b64 exec, exec, vcc\n\t"
      "s_cbranch_execz 2f\n\t"
      "1:\n\t"
      "v_add_u32 v74, %[st], %[lnk]\n\t"
      "ds_read_b128 v[64:67], %[st]\n\t"              // centre.xyz, half.x
      "ds_read_b64 v[68:69], %[st] offset:16\n\t"     // half.yz
      "ds_read_b32 %[lw], v74\n\t"                     // hit | miss << 16 for this ray's octant
      "s_waitcnt lgkmcnt(1)\n\t"
      // slab distances from the box's centre and half extent: t(centre) -+ half * |1/d| are the entry and exit distances of
      // an axis whatever the sign of d — nine fma and no min / max per axis (the lo / hi form took six fma and six min / max)
      "v_fma_f32 v64, v64, %[ix], %[nx]\n\t"
      "v_fma_f32 v65, v65, %[iy], %[ny]\n\t"
      "v_fma_f32 v66, v66, %[iz], %[nz]\n\t"
      "v_fma_f32 v74, -v67, |%[ix]|, v64\n\t"
      "v_fma_f32 v64, v67, |%[ix]|, v64\n\t"
      "v_fma_f32 v67, -v68, |%[iy]|, v65\n\t"
      "v_fma_f32 v65, v68, |%[iy]|, v65\n\t"
      "v_fma_f32 v68, -v69, |%[iz]|, v66 clamp\n\t"   // max(entry z, 0); the upper clamp at 1 = 2^17 lies beyond MAX_DIST
      "v_fma_f32 v66, v69, |%[iz]|, v66\n\t"
      "v_max3_f32 v74, v74, v67, v68\n\t"              // tnear
      "v_min3_f32 v64, v64, v65, v66\n\t"              // tfar
      // hit <=> tnear <= tfar && 0 <= tfar && tnear <= best  <=>  max(tnear, 0) <= min(tfar, best)   (best >= 0);
      // the max with 0 came with the z entry distance's clamp (a box entered beyond 2^17 reads as entered AT 2^17: it can only
      // pass while no hit is known, and then it is one visit too many, never one too few)
      "v_min_f32 v64, v64, %[best]\n\t"
      "v_cmp_le_f32 vcc, v74, v64\n\t"                 // box hit
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_nop 0\n\t"                                    // a VALU read of VCC needs 2 wait states after the VALU compare that wrote it
      "v_cndmask_b32_sdwa %[st], %[lw], %[lw], vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0\n\t"
      // a node address: keep walking (else parked at a leaf, or done).  The compare writes EXEC itself: one scalar instruction less per
      // iteration than v_cmp + s_and_b64 — the CU's one scalar pipe issues half as many instructions per clock as its four vector pipes
      // (scripts/ubench/ifetch_rate.hip) and this kernel keeps both ~58 % busy: +0.7 % on the headline
      "v_cmpx_gt_u32 vcc, 0x8000, %[st]\n\t"
      "s_bcnt1_i32_b64 %[walkers], exec\n\t"
      "s_cmp_ge_u32 %[walkers], %[wmin]\n\t"
      "s_cbranch_scc1 1b\n\t"
      "2:\n\t"
      "s_mov_b64 exec, %[save]\n\t"
#ifdef PT_WALK_PRIO
      "s_setprio 0\n\t"
#endif
      : [st] "+v"(state), [lw] "=&v"(link_word), [save] "=&s"(save), [walkers] "=&s"(walkers)
      : [wmin] "s"(walk_min), [lnk] "v"(lnk), [ix] "v"(w.inv.x), [iy] "v"(w.inv.y), [iz] "v"(w.inv.z),
        [nx] "v"(w.noi.x), [ny] "v"(w.noi.y), [nz] "v"(w.noi.z), [best] "v"(w.best.t * PT_T_SCALE)
      : "v64", "v65", "v66", "v67", "v68", "v69", "v74", "vcc", "scc", "memory");
  leaf_first = 0u;
  leaf_count = 0u;
  if (state >= 0x8000u && state != 0xFFFFu) {               // parked: 0x8000 | count << 11 | first triangle record
    leaf_first = state & 0x7FFu;
    leaf_count = (state >> 11) & 0xFu;
    state = link_word >> 16;                                // the leaf's miss code: where the walk goes on after the tests
  }
}

// from a node's box to its link word for this ray's octant
PT_DEV uint32_t walk_lds_link_offset(const Walk& w) { return w.n_nodes * 32u + w.oct * 4u; }
PT_DEV uint32_t walk_lds_state(uint32_t lds_nodes, uint32_t node) { return node == PT_END ? 0xFFFFu : lds_nodes + (node << 5); }
PT_DEV uint32_t walk_lds_node(uint32_t lds_nodes, uint32_t state) { return state == 0xFFFFu ? PT_END : (state - lds_nodes) >> 5; }

// one box phase in terms of node indices (the tile / persistent / split kernels)
PT_DEV void walk_to_leaf_lds(uint32_t lds_nodes, Walk& w, uint32_t& leaf_first, uint32_t& leaf_count, uint32_t walk_min = 1u)
{
  uint32_t state = walk_lds_state(lds_nodes, w.node);
  walk_to_leaf_lds_state(state, walk_lds_link_offset(w), w, leaf_first, leaf_count, walk_min);
  w.node = walk_lds_node(lds_nodes, state);
}

template <bool STATS, bool ASM = false>
PT_DEV void walk_leaf(const float4* tris, Walk& w, uint32_t leaf_first, uint32_t leaf_count, uint32_t& n_tris,
                      uint32_t& wave_tri_iters, bool small_det = false, uint32_t stride = 3u /* float4 per triangle record */)
{
  if (ASM && !STATS && PT_ASM_LEAF && small_det) {   // (small_det is wave-uniform: one scalar branch per leaf phase)
    for (uint32_t k = 0; k < leaf_count; ++k) {
      const float4* r = tris + (leaf_first + k) * stride;
      const float4 a = r[0], b = r[1];
      const float2 c = *reinterpret_cast<const float2*>(r + 2);
      mt_test_asm(a, b, c, w.o, w.d, w.best);
    }
    return;
  }
  for (uint32_t k = 0; k < leaf_count; ++k) {
    if (STATS) {
      const unsigned long long act = __ballot(1);
      if ((threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)act) - 1)) ++wave_tri_iters;
    }
    const uint32_t ti = (leaf_first + k) * stride;
    mt_test<false>(tris[ti], tris[ti + 1], tris[ti + 2], w.o, w.d, w.best, small_det);
  }
  if (STATS) n_tris += leaf_count;
}

// "while-while": all lanes walk boxes until each holds a leaf (or is done), then all lanes run
// Moller-Trumbore on their leaf, repeat.
template <bool STATS, bool NODES_IN_LDS = false>
PT_DEV void traverse_bvh(const float4* nodes, const float4* tris, uint32_t n_nodes, f3 o, f3 d,
                         Best& best, uint32_t& n_nodes_visited, uint32_t& n_tris, uint32_t& wave_node_iters,
                         uint32_t& wave_tri_iters, uint32_t* idle3)
{
  Walk w;
  walk_init(w, o, d, n_nodes, NODES_IN_LDS && !STATS && PT_ASM_WALK);
  if (STATS) w.alive = (uint32_t)__popcll(__ballot(1));
  // LDS byte address of the node table (low 32 bits of the flat address of an LDS object)
  const uint32_t lds_nodes = (uint32_t)(uintptr_t)nodes;
  for (;;) {
    uint32_t leaf_first, leaf_count;
    if (NODES_IN_LDS && !STATS && PT_ASM_WALK) walk_to_leaf_lds(lds_nodes, w, leaf_first, leaf_count);
    else walk_to_leaf<STATS>(nodes, w, leaf_first, leaf_count, n_nodes_visited, wave_node_iters);
    // a box phase can end early (PT_LEAF_MIN) with this lane still walking: no leaf and not at the end
    if (leaf_count != 0u) walk_leaf<STATS>(tris, w, leaf_first, leaf_count, n_tris, wave_tri_iters);
    else if (w.node == PT_END) break;
  }
  best = w.best;
  if (STATS) { idle3[0] += w.idle_unstarted; idle3[1] += w.idle_finished; idle3[2] += w.idle_parked; }
}

// intersection.cuh:140-155 (see the oracle's note on the discarded conditional at :152)
PT_DEV bool intersect_sphere(f3 o, f3 d, f3 center, float radius2 /* radius * radius, formed by the host */, float& t)
{
  const float epsilon = 0.01f;
  const f3 op = center - o;
  const float b = dot(op, d);
  float disc = b * b - dot(op, op) + radius2;
  if (disc < 0.0f) return false;
  disc = sqrt_checked(disc);
  t = b - disc;
  if (!(t > epsilon)) t = b + disc;
  return t != 0.0f;
}

// intersection.cuh:20-26
PT_DEV int texture_idx(const TexDesc& tex, float uvx, float uvy)
{
  int x = (int)(uvx * (float)(tex.w - 1));
  int y = (int)(uvy * (float)(tex.h - 1));
  return (y * tex.w + x) * tex.nb_chan;
}

struct Counters { uint32_t rays, nodes, tris, mesh_hits, nmap_hits, wave_node_iters, wave_tri_iters, fetch_events, fetch_rays, idle3[3];
                  // instrumented restart kernel: shader-clock cycles of a wave by phase — 0 pool refill, 1 box phases, 2 leaf phases, 3 light loop + shading,
                  // 4 the whole round loop; 5 leaf phases entered; 8 shading record fetch; 9 path_post + parking; 10 path_post up to the BSDF sample; 11 the BSDF sample
                  // (added by one lane per wave: sums over waves)
                  unsigned long long cyc[16]; };   // [6] node fetch (issue -> data), [7] box tests + pushes, [8] pops after a visit without a hit (instrumented four-wide float walk)
// one lane per executing wave adds a wave-level measurement
#define PT_WAVE_ONE() ((threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)__ballot(1)) - 1))

// Result of the nearest-hit search of one intersect() call, before any shading data is touched:
// 16 bytes, which is what travels through LDS when rays are compacted across a workgroup.
// idx: global face index; PT_END = nothing; PT_LIGHT | l = light sphere l won.
struct Nearest { float t, u, v; uint32_t idx; };
#define PT_LIGHT 0x80000000u

// The light loop of intersect() (intersection.cuh:199-212) applied to the face search's result.
// The light table is read-only for the whole launch and indexed wave-uniformly here: read it through the constant
// address space so the loads become s_load_dwordx4 (scalar cache, SGPR results) instead of per-lane VMEM loads with a
// vmcnt(0) stall per light.
typedef const __attribute__((address_space(4))) float* ConstF;
PT_DEV ConstF as_constant(const float4* q) { return (ConstF)(uintptr_t)q; }

// Launch constants that only the rare parts of a persistent kernel need (a tile's prologue, a finished path, a miss) are
// re-read from the kernel-argument segment where they are used — KParams is every kernel's first argument — behind a
// compiler barrier, instead of living in SGPRs for the whole launch: the round loop is short of scalar registers
// (43 SGPR spills = v_writelane / v_readlane in the loop before this, 8.10 -> 8.33 Gsamples/s for the camera alone).
// PT_KARG reads KParams fields from the start of the kernel-argument segment WHATEVER object it is handed: every kernel whose
// device functions use it must take the launch constants as its FIRST by-value argument — which is what this macro spells, so
// that a kernel cannot be written otherwise by accident.
#define PT_KERNEL_PARAMS const KParams p
#if PT_CAM_FROM_KERNARG
template <typename T>
PT_DEV T karg_load(uint32_t byte_offset)
{
  // the barrier sits on the segment pointer, not on the field's address: every use is then `s_load ..., base, offset` off
  // ONE scalar pair instead of a hoisted (and spilled) address pair per field
  const __attribute__((address_space(4))) char* q = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(q));
  return *(const __attribute__((address_space(4))) T*)(q + byte_offset);
}
#ifndef PT_KARG_ALL
#define PT_KARG_ALL 1
#endif
#if PT_KARG_ALL
#define PT_KARG(p, field) karg_load<decltype(KParams::field)>((uint32_t)__builtin_offsetof(KParams, field))
#else
#define PT_KARG(p, field) ((p).field)
#endif
#else
#define PT_KARG(p, field) ((p).field)
#endif

PT_DEV Nearest nearest_lights(const KParams& p, f3 o, f3 d, Nearest n)
{
  const ConstF lights = as_constant(p.lights);     // 8 floats per light: color.xyz, vec.xyz, emission, radius^2 (ptamd_api.cpp)
  for (uint32_t l = 0; l < p.n_lights; ++l) {
    const ConstF L = lights + l * 8u;
    // the discriminant of intersect_sphere, same operations: when it is negative for every lane of the wave (the usual
    // case for a small light) the correctly-rounded square root and everything after it is skipped for the whole wave
    const f3 c = mk3(L[3], L[4], L[5]);
    const f3 op = c - o;
    const float b = dot(op, d);
    const float disc = b * b - dot(op, op) + L[7];
    if (__builtin_amdgcn_ballot_w64(!(disc < 0.0f)) == 0ull) continue;
    float t;
    if (intersect_sphere(o, d, c, L[7], t) && t < n.t && t >= 0.0f) {
      n.t = t;
      n.idx = PT_LIGHT | l;
    }
  }
  return n;
}

// First half of intersection.cuh:161-246: the two search loops (faces :179-196, lights :199-212).
// KIND: 1 brute force, 2 BVH.
template <int KIND, bool STATS, bool NODES_IN_LDS = false>
PT_DEV Nearest trace_nearest(const KParams& p, const float4* s_nodes, const float4* s_tris, f3 o, f3 d,
                             Counters& cnt)
{
  Best best;
  best.t = PT_MAX_DIST; best.u = 0.f; best.v = 0.f; best.idx = PT_END;
  if (STATS) cnt.rays++;
  if (KIND == 1) traverse_brute<STATS>(s_tris, p.n_faces, o, d, best, cnt.tris);
  else traverse_bvh<STATS, NODES_IN_LDS>(s_nodes, s_tris, p.n_nodes, o, d, best, cnt.nodes, cnt.tris, cnt.wave_node_iters, cnt.wave_tri_iters, cnt.idle3);
  Nearest n;
  n.t = best.t; n.u = best.u; n.v = best.v; n.idx = best.idx;
  return nearest_lights(p, o, d, n);
}

// Second half of intersection.cuh:161-246 for the winner only: the deferred interpolation of
// intersectTriangle (:124-131), the light normal (:204-210), texture and normal-map fetches
// (:216-243).  Returns intersection.dist < MAX_DIST.
template <bool STATS>
PT_DEV bool resolve_hit(const KParams& p, f3 d, Nearest n, Hit& hit, Counters& cnt)
{
  hit.dist = n.t;
  if (n.idx == PT_END) return false; // n.t == MAX_DIST
  if (n.idx & PT_LIGHT) {
    const uint32_t l = n.idx & ~PT_LIGHT;
    const float4 la = p.lights[l * 2 + 0], lb = p.lights[l * 2 + 1];
    hit.light = (int)l;
    hit.emission = lb.z;
    hit.diffuse_col = mk3(la.x, la.y, la.z);
    hit.normal = normalize_hot(mk3(la.w, lb.x, lb.y) - (n.t * d)); // intersection.cuh:208: origin ignored
    return hit.dist < PT_MAX_DIST;
  }
  // one burst of six independent 16-byte loads: geometry, material and the diffuse+specular map of the face (the
  // normal map's descriptor follows only for normal-mapped materials)
  const float4* sh = p.shade + (size_t)n.idx * 7;
  const float4 s0 = sh[0], s1 = sh[1], s2 = sh[2], s3 = sh[3], s4 = sh[4], s5 = sh[5];
  const f3 n0 = mk3(s0.x, s0.y, s0.z), n1 = mk3(s0.w, s1.x, s1.y), n2 = mk3(s1.z, s1.w, s2.x);
  const float u = n.u, v = n.v;
  const float w = 1.0f - u - v;
  const f3 surface_normal = w * n0 + u * n1 + v * n2;
  // texture coordinates: only faces with a real texture or a normal map need them (none of indoor.obj's do), so they are
  // computed where they are used
  auto tex_uv = [&](float& uvx, float& uvy) {
    const float ux = w * s2.y + u * s2.w + v * s3.y;
    const float uy = w * s2.z + u * s3.x + v * s3.z;
    uvx = ux - __builtin_floorf(ux / 1.0f); // mod(uv, 1.0): cutils_math.h:1728-1737
    uvy = uy - __builtin_floorf(uy / 1.0f);
  };
  const f3 tangent = mk3(s3.w, s4.x, s4.y);
  hit.normal = surface_normal;
  hit.light = -1;
  hit.ior = s4.w;
  if (f_as_u(s4.z) & 0x80000000u) {
    // 1x1 diffuse+specular map: the record carries its only texel (ptamd_api.cpp) — no dependent load
    hit.diffuse_col = mk3(s5.x, s5.y, s5.z);
    hit.specular_col = s5.w;
  } else {
    TexDesc tex;
    tex.w = (int32_t)f_as_u(s5.x); tex.h = (int32_t)f_as_u(s5.y); tex.nb_chan = (int32_t)f_as_u(s5.z); tex.pad = 0; tex.offset = f_as_u(s5.w);
    float uvx, uvy;
    tex_uv(uvx, uvy);
    const float* texel = PT_KARG(p, texels) + tex.offset + texture_idx(tex, uvx, uvy);
    hit.diffuse_col = mk3(texel[0], texel[1], texel[2]);
    hit.specular_col = texel[3];
  }
  if (STATS) cnt.mesh_hits++;
  if (f_as_u(s4.z) & 0x40000000u) {   // normal-mapped material (flag set at upload): its descriptor is the record's 7th float4
    const float4 s6 = sh[6];
    TexDesc nt;
    nt.w = (int32_t)f_as_u(s6.x); nt.h = (int32_t)f_as_u(s6.y); nt.nb_chan = (int32_t)f_as_u(s6.z); nt.pad = 0; nt.offset = f_as_u(s6.w);
    float uvx, uvy;
    tex_uv(uvx, uvy);
    const float* nx = PT_KARG(p, texels) + nt.offset + texture_idx(nt, uvx, uvy);
    const f3 nn = normalize((mk3(nx[0], nx[1], nx[2]) * 2.0f) - 1.0f);
    const f3 binormal = normalize(cross(tangent, surface_normal));
    const f3 tx = tangent, ty = -binormal, tz = surface_normal;
    hit.normal = mk3(tx.x * nn.x + ty.x * nn.y + tz.x * nn.z,
                     tx.y * nn.x + ty.y * nn.y + tz.y * nn.z,
                     tx.z * nn.x + ty.z * nn.y + tz.z * nn.z);
    if (STATS) cnt.nmap_hits++;
  }
  return hit.dist < PT_MAX_DIST;
}

// texCubemap restated (see oracle/pt_oracle.c: or_tex_cubemap for the definition)
PT_DEV f3 env_lookup(const KParams& p, f3 dir)
{
  // one-colour environment (flag set by the host, ptamd_upload_cubemap): the face choice below cannot matter
  if (PT_KARG(p, env_uniform)) return mk3(PT_KARG(p, env_r), PT_KARG(p, env_g), PT_KARG(p, env_b));
  float x = dir.x, y = dir.y, z = -dir.z; // raytrace.cu:60,197
  // (the compiler otherwise computes |dir| where the walk is set up, for both uses, and spills it across the walk)
  asm volatile("" : "+v"(x), "+v"(y), "+v"(z));
  const uint32_t n = PT_KARG(p, cubemap_size);
  const float4* cubemap = PT_KARG(p, cubemap);
  const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y), az = __builtin_fabsf(z);
  int face;
  float m, s, t;
  if (ax >= ay && ax >= az) {
    m = ax;
    if (x >= 0.0f) { face = 0; s = -z; t = -y; } else { face = 1; s = z; t = -y; }
  } else if (ay >= az) {
    m = ay;
    if (y >= 0.0f) { face = 2; s = x; t = z; } else { face = 3; s = x; t = -z; }
  } else {
    m = az;
    if (z >= 0.0f) { face = 4; s = x; t = -y; } else { face = 5; s = -x; t = -y; }
  }
  const float4* base = cubemap + (size_t)face * n * n;
  if (n == 1) { const float4 c = base[0]; return mk3(c.x, c.y, c.z); }
  const float u = (s / m + 1.0f) * 0.5f;
  const float v = (t / m + 1.0f) * 0.5f;
  const float xb = u * (float)n - 0.5f;
  const float yb = v * (float)n - 0.5f;
  const float fx = __builtin_floorf(xb), fy = __builtin_floorf(yb);
  float a = __builtin_floorf((xb - fx) * 256.0f) * (1.0f / 256.0f);
  float b = __builtin_floorf((yb - fy) * 256.0f) * (1.0f / 256.0f);
  int i0 = (int)fx, j0 = (int)fy, i1 = i0 + 1, j1 = j0 + 1;
  const int hi = (int)n - 1;
  if (!(xb == xb)) { i0 = i1 = 0; a = 0.0f; }
  if (!(yb == yb)) { j0 = j1 = 0; b = 0.0f; }
  i0 = i0 < 0 ? 0 : (i0 > hi ? hi : i0);
  i1 = i1 < 0 ? 0 : (i1 > hi ? hi : i1);
  j0 = j0 < 0 ? 0 : (j0 > hi ? hi : j0);
  j1 = j1 < 0 ? 0 : (j1 > hi ? hi : j1);
  const float4 t00 = base[(size_t)j0 * n + i0], t10 = base[(size_t)j0 * n + i1];
  const float4 t01 = base[(size_t)j1 * n + i0], t11 = base[(size_t)j1 * n + i1];
  const f3 top = mk3(t00.x, t00.y, t00.z) * (1.0f - a) + mk3(t10.x, t10.y, t10.z) * a;
  const f3 bot = mk3(t01.x, t01.y, t01.z) * (1.0f - a) + mk3(t11.x, t11.y, t11.z) * a;
  return top * (1.0f - b) + bot * b;
}

// ---------------------------------------------------------------- one path = one sample of one pixel

// Per-lane state of a path between two iterations of the reference's bounce loop
// (raytrace.cu:67).  Holding it in a struct lets lanes of one wave sit at different bounce
// depths, which is what the persistent kernel's refill relies on.
struct Path {
  f3 o, d, throughput, acc;
  Xorwow rng;
  float specular_col; // carried over between iterations
  uint32_t xy; // pixel: x | y << 16 (frames are at most 65536 wide/high, checked by the API)
  uint32_t bk; // bounce index b | sample index k << 16 (k: frame inside a batched launch, 0 otherwise)
  PT_DEV uint32_t x() const { return xy & 0xffffu; }
  PT_DEV uint32_t y() const { return xy >> 16; }
  PT_DEV uint32_t b() const { return bk & 0xffffu; }
  PT_DEV uint32_t k() const { return bk >> 16; }
};

// kernel() prologue: seed, generateRay, camera_dof (raytrace.cu:227-240)
PT_DEV void path_begin(const KParams& p, uint32_t x, uint32_t y, Path& st, uint32_t k = 0)
{
  // raytrace.cu:227-229 with the reference's launch geometry (16x16 blocks, padded grid)
  const uint32_t width = PT_KARG(p, width);
  const uint32_t grid_x = width / 16u + 1u;
  const uint32_t tid = ((x >> 4) + (y >> 4) * grid_x) * 256u + (y & 15u) * 16u + (x & 15u);
  // k > 0 only in batched launches: frame frame_nb0 + k has hash_seed WangHash(frame_nb0 + k) (raytrace.cu:321)
  const uint32_t hash_seed = k == 0 ? PT_KARG(p, hash_seed) : wang_hash(PT_KARG(p, frame_nb0) + k);
  xorwow_init(st.rng, hash_seed + tid);

  // The 14 camera constants are only needed here, once per tile in the persistent kernels: they are re-read from the
  // kernel-argument segment (KParams is every kernel's first argument) behind a compiler barrier instead of being kept
  // in SGPRs for the whole launch, where they cost the round loop a dozen scalar spills (v_writelane / v_readlane).
#if PT_CAM_FROM_KERNARG
  ConstF cam = (ConstF)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + __builtin_offsetof(KParams, cam_pos));
  asm volatile("" : "+s"(cam));
  const f3 cam_pos = mk3(cam[0], cam[1], cam[2]), cam_p0 = mk3(cam[3], cam[4], cam[5]);
  const f3 cam_u = mk3(cam[6], cam[7], cam[8]), cam_v = mk3(cam[9], cam[10], cam[11]);
  const float focus_dist = cam[12], aperture = cam[13];
#else
  const f3 cam_pos = p.cam_pos, cam_p0 = p.cam_p0, cam_u = p.cam_u, cam_v = p.cam_v;
  const float focus_dist = p.focus_dist, aperture = p.aperture;
#endif
  // generateRay (intersection.cuh:75-97), pixel-invariant terms precomputed on the host
  const int half_w = (int)(width / 2u), half_h = (int)(PT_KARG(p, height) / 2u);
  const f3 screen_pos = (cam_p0 + (cam_u * (float)((int)x - half_w))) + (cam_v * (float)((int)y - half_h));
  f3 dir = normalize_hot(screen_pos - cam_pos);
  f3 origin = cam_pos;

  // camera_dof (post_process.cuh:49-67)
  const f3 focal_point = focus_dist * dir;
  const float random_angle = (float)((double)(xorwow_uniform(st.rng) * 2.0f) * 3.14159265358979323846);
  const float random_radius = xorwow_uniform(st.rng) * aperture;
  float sn, cs;
  pt_sincosf(random_angle, sn, cs);
  const f3 ap = (cs * cam_u + sn * cam_v) * random_radius;
  st.d = normalize_hot(focal_point - ap);
  st.o = origin + ap;

  st.throughput = mk3(1.0f);
  st.acc = mk3(0.0f);
  st.specular_col = 0.0f;
  st.xy = x | (y << 16);
  st.bk = k << 16;
}

// One iteration of radiance()'s loop (raytrace.cu:67-207) is split around the nearest-hit search
// so that the search can run in another lane (workgroup ray compaction):
//   r1 = path_pre(st);  n = trace_nearest(st.o, st.d);  done = path_post(st, r1, n);
// When the camera moved, the "iteration" is the whole preview sample (raytrace.cu:54-62).
PT_DEV float path_pre(const KParams& p, Path& st)
{
  return p.is_static ? xorwow_uniform(st.rng) : 0.0f; // raytrace.cu:70: drawn before tracing
}

// Returns true when the path is complete (st.acc is final).
template <bool STATS>
PT_DEV bool path_post(const KParams& p, Path& st, float r1, Nearest nearest, Counters& cnt)
{
  Hit inter;
  inter.normal = mk3(0.f); inter.diffuse_col = mk3(0.f);
  inter.dist = 0.f; inter.specular_col = st.specular_col; inter.ior = 0.f; inter.light = -1; inter.emission = 0.f;
  long long ts0 = 0;
  if (STATS) ts0 = clock64();
  const bool found = resolve_hit<STATS>(p, st.d, nearest, inter, cnt);
  if (STATS) {   // [8]: the shading record (and texels) of the hit: fetch and decode
    asm volatile("" : "+v"(inter.normal.x), "+v"(inter.diffuse_col.x), "+v"(inter.specular_col));
    const long long ts1 = clock64();
    if (PT_WAVE_ONE()) cnt.cyc[8] += (unsigned long long)(ts1 - ts0);
  }

  if (!p.is_static) {
    st.acc = found ? inter.diffuse_col : env_lookup(p, st.d);
    return true;
  }

  const int max_bounces = p.bounces;
  st.specular_col = inter.specular_col;
  if (!found) {
    // The reference keeps looping with the unchanged ray (raytrace.cu:194-199): every remaining
    // iteration misses again and adds the same environment sample.  Run them here, without
    // the walk, drawing r1 exactly as the loop does.
    const f3 env = env_lookup(p, st.d);
    for (;;) {
      st.acc = st.acc + env * st.throughput;
      const float pmax = __builtin_fmaxf(st.throughput.x, __builtin_fmaxf(st.throughput.y, st.throughput.z));
      if (r1 > pmax && st.b() > 1u) return true;
      st.throughput = st.throughput * rcp_hot(pmax);
      ++st.bk;
      if ((int)st.b() >= max_bounces) return true;
      r1 = xorwow_uniform(st.rng);
      if (STATS) cnt.rays++; // the reference issues this (futile) intersect() call; count it as a ray
    }
  }

  long long tp0 = 0;
  if (STATS) { tp0 = clock64(); if (PT_WAVE_ONE()) cnt.cyc[10] += (unsigned long long)(tp0 - ts0); }   // [10]: up to here (fetch + the misses' loop)
  const f3 d = st.d;
  const float cos_theta = dot(inter.normal, d);
  f3 oriented_normal = inter.normal;
  const f3 spec = normalize_hot(reflect(d, inter.normal));
  const f3 direct_light = inter.diffuse_col / 0.5f; // brdf_lambert / pdf_lambert (brdf.cuh:14-31)
  if (inter.ior == 1.0f || inter.light >= 0) {
    if (inter.light >= 0) {
      // raytrace.cu:86: (light colour * emission) * throughput; resolve_hit has both in registers (no second fetch of the record)
      st.acc = st.acc + (inter.diffuse_col * inter.emission) * st.throughput;
    }
    const float phi = (float)((double)2.0f * 3.14159265358979323846 * (double)xorwow_uniform(st.rng));
    // r1 is a uniform variate in (0, 1] (or 0): both arguments are 0 or >= 2^-33
    const float sin_t = sqrt_hot(r1);
    const float cos_t = sqrt_hot(1.f - r1);
    const f3 axis = (__builtin_fabsf(oriented_normal.x) >= 0.1f) /* == (double)|x| > .1: 0.1f is the first float above the double .1 */ ? mk3(0.0f, 1.0f, 0.0f) : mk3(1.0f, 0.0f, 0.0f);
    const f3 u = normalize_hot(cross(axis, oriented_normal));
    const f3 v = cross(oriented_normal, u);
    float sphi, cphi;
    pt_sincosf(phi, sphi, cphi);
    const f3 dd = normalize_hot(v * sin_t * cphi + u * sphi * sin_t + oriented_normal * cos_t);
    st.o = st.o + d * inter.dist;
    st.d = mix(dd, spec, inter.specular_col);
    st.o = st.o + st.d * 0.03f;
    st.throughput = st.throughput * direct_light;
  } else {
    const float n1 = 1.0f;
    const float n2 = inter.ior;
    oriented_normal = cos_theta < 0 ? inter.normal : inter.normal * -1.0f;
    const float c1 = dot(oriented_normal, d);
    const bool entering = dot(inter.normal, oriented_normal) > 0;
    const float eta = entering ? n1 / n2 : n2 / n1;
    const float eta_2 = eta * eta;
    const float c2_term = 1.0f - eta_2 * (1.0f - c1 * c1);
    if (c2_term < 0.0f) {
      st.o = st.o + oriented_normal * inter.dist / 100.f;
      st.d = spec;
    } else {
      float R0 = (n2 - n1) / (n1 + n2);
      R0 *= R0;
      const float c2 = __builtin_sqrtf(c2_term);
      const f3 T = normalize(eta * d + (eta * c1 - c2) * oriented_normal);
      const float f_cos_theta = pt_powf(cos_theta, 5.0f);
      const float f_r = R0 + (1.0f - R0) * f_cos_theta;
      if (xorwow_uniform(st.rng) < 0.25f) {
        st.throughput = st.throughput * (f_r * direct_light);
        st.o = st.o + oriented_normal * inter.dist / 100.f;
        st.d = spec;
      } else {
        const float f_t = 1.0f - f_r;
        st.throughput = st.throughput * (f_t * direct_light);
        st.o = st.o + oriented_normal * inter.dist / 10000.f;
        st.d = T;
      }
    }
  }
  if (STATS) { asm volatile("" : "+v"(st.d.x), "+v"(st.o.x), "+v"(st.throughput.x)); const long long tp1 = clock64(); if (PT_WAVE_ONE()) cnt.cyc[11] += (unsigned long long)(tp1 - tp0); }   // [11]: the BSDF sample
  const float pmax = __builtin_fmaxf(st.throughput.x, __builtin_fmaxf(st.throughput.y, st.throughput.z));
  if (r1 > pmax && st.b() > 1u) return true;
  st.throughput = st.throughput * rcp_hot(pmax);
  ++st.bk;
  return (int)st.b() >= max_bounces;
}

// ---------------------------------------------------------------- post process

PT_DEV f3 uncharted_tonemap(f3 x) // post_process.cuh:14-25
{
  const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
  return ((x * (A * x + C * B) + D * E) / (x * (A * x + B) + D * F)) - E / F;
}

PT_DEV f3 exposure(f3 color) // post_process.cuh:31-41
{
  const float exposure_bias = 2.0f;
  const f3 curr = uncharted_tonemap(exposure_bias * color);
  const f3 W = mk3(11.2f);
  const f3 white_scale = 1.0f / uncharted_tonemap(W);
  return curr * white_scale;
}

PT_DEV f3 post_process(uint32_t id, f3 c) // raytrace.cu:327-352
{
  if (id == 1) {
    const float gray = (float)((double)c.x * 0.3 + (double)c.y * 0.59 + (double)c.z * 0.11);
    return mk3(gray, gray, gray);
  }
  if (id == 2)
    return mk3((float)((double)c.x * 0.393 + (double)c.y * 0.769 + (double)c.z * 0.189),
               (float)((double)c.x * 0.349 + (double)c.y * 0.686 + (double)c.z * 0.168),
               (float)((double)c.x * 0.272 + (double)c.y * 0.534 + (double)c.z * 0.131));
  if (id == 3)
    return mk3((float)(1.0 - (double)c.x), (float)(1.0 - (double)c.y), (float)(1.0 - (double)c.z));
  return c;
}

// ---------------------------------------------------------------- LDS staging

// Copies n16 16-byte words global -> LDS with the whole workgroup, coalesced.
PT_DEV void stage_to_lds(float4* dst, const float4* src, uint32_t n16)
{
  for (uint32_t i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i];
}

// The unsplit form used when a lane traces its own ray.
template <int KIND, bool STATS, bool NODES_IN_LDS = false>
PT_DEV bool path_step(const KParams& p, const float4* s_nodes, const float4* s_tris, Path& st, Counters& cnt)
{
  const float r1 = path_pre(p, st);
  const Nearest n = trace_nearest<KIND, STATS, NODES_IN_LDS>(p, s_nodes, s_tris, st.o, st.d, cnt);
  return path_post<STATS>(p, st, r1, n, cnt);
}

// ---------------------------------------------------------------- the gamma step of the tonemap as a table
//
// raytrace.cu:262-268: colour = pow(colour, 1/2.2), then each channel * 255 is stored through a truncating conversion.  With
// no post-process in between, the stored byte is a monotone step function of the value going into pow, and pt_powf — the
// build's binary64 definition of powf, ~70 instructions, 50 of them f64 — is by far the dearest thing the resolve pass does
// (three per pixel).  The 256 steps are found once per context with pt_powf itself (pt_build_gamma_table, a binary search
// over bit patterns per step) and the byte is read off them: a 3-instruction estimate (v_log, v_mul, v_exp) that is
// within one step of the answer, corrected by two comparisons against the table.  ptamd_gamma_table_selftest compares the
// result with the pt_powf form for EVERY binary32 value below T[256] on the device.
PT_DEV uint32_t gamma_byte_exact(float x)   // the definition (post_id 0): what the reference's store sequence produces
{
  return pt_f2u(pt_powf(x, 1.0f / 2.2f) * 255.0f) & 0xffu;
}
PT_DEV uint32_t gamma_value_exact(float x) { return pt_f2u(pt_powf(x, 1.0f / 2.2f) * 255.0f); }

// T points at 258 floats (LDS copy in the resolve kernels).  Valid for every x: values at or above T[256], and NaN, take the
// pt_powf form; negative values and zero give 0 either way (pt_powf: NaN or 0 -> pt_f2u -> 0).
template <typename TablePtr>
PT_DEV uint32_t gamma_byte(float x, TablePtr T)
{
  if (!(x < T[256])) return gamma_byte_exact(x);
  if (!(x > 0.0f)) return 0u;
  const float est = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(x) * (1.0f / 2.2f)) * 255.0f;
  int k = (int)est;
  k = k < 0 ? 0 : (k > 255 ? 255 : k);
  k = x < T[k] ? k - 1 : k;          // T[0] = 0 < x: k stays >= 0
  k = x >= T[k + 1] ? k + 1 : k;     // x < T[256]: k stays <= 255
  return (uint32_t)k;
}

// kernel() epilogue: clamp, temporal accumulation, tonemap, gamma, post-process, RGBA8 store
// (raytrace.cu:248-270)
PT_DEV void path_finish(const KParams& p, const Path& st)
{
  f3 rad = mk3(clamp01(st.acc.x), clamp01(st.acc.y), clamp01(st.acc.z));
  const uint32_t x = st.x(), y = st.y();
  // row-flipped accumulator index (raytrace.cu:252)
  const size_t i = (size_t)(p.height - y - 1u - p.tfb_row0) * p.width + x;
  float* tp = p.tfb + i * 3;
  f3 t = p.tfb_reset ? mk3(0.0f) : mk3(tp[0], tp[1], tp[2]);   // tfb_reset: the accumulator counts as freshly zeroed
  t = t * (float)p.is_static;
  t = t + rad;
  tp[0] = t.x; tp[1] = t.y; tp[2] = t.z;
  rad = p.frame_nb_inv != 0.0f ? t * p.frame_nb_inv : t / p.frame_nb_f;
  rad = exposure(rad);
  uint32_t px;
  if (p.gamma_table && p.post_id == 0u) {   // the byte read off the gamma table (two loads per channel) instead of three pt_powf
    px = gamma_byte(rad.x, p.gamma_table) | (gamma_byte(rad.y, p.gamma_table) << 8) | (gamma_byte(rad.z, p.gamma_table) << 16);
  } else {
    const float g = 1.0f / 2.2f;
    rad = mk3(pt_powf(rad.x, g), pt_powf(rad.y, g), pt_powf(rad.z, g));
    rad = post_process(p.post_id, rad);
    px = (pt_f2u(rad.x * 255.0f) & 0xffu) | ((pt_f2u(rad.y * 255.0f) & 0xffu) << 8) | ((pt_f2u(rad.z * 255.0f) & 0xffu) << 16);
  }
  p.surface[(size_t)(y - p.surf_row0) * p.width + x] = px;
}

// Batched launches: the clamped sample (raytrace.cu:248) is parked; pt_resolve_kernel applies the
// samples of a pixel to the accumulator in frame order, which is what consecutive launches do.
PT_DEV void path_finish_sample(const KParams& p, const Path& st)
{
  const uint32_t row_begin = PT_KARG(p, row_begin), rows = PT_KARG(p, row_end) - row_begin;
  float* sp = PT_KARG(p, samples_out) + (((size_t)st.k() * rows + (st.y() - row_begin)) * PT_KARG(p, width) + st.x()) * 3;
  sp[0] = clamp01(st.acc.x); sp[1] = clamp01(st.acc.y); sp[2] = clamp01(st.acc.z);
}

template <bool STATS>
PT_DEV void flush_counters(const KParams& p, const Counters& cnt, uint32_t samples)
{
  if (!STATS) return;
  const uint32_t lane = threadIdx.x & 63u;
  unsigned long long v[13] = { cnt.rays, cnt.nodes, cnt.tris, cnt.mesh_hits, cnt.nmap_hits, samples,
                               cnt.wave_node_iters, cnt.wave_tri_iters, cnt.fetch_events, cnt.fetch_rays,
                               cnt.idle3[0], cnt.idle3[1], cnt.idle3[2] };
  for (int k = 0; k < 13; ++k) {
    unsigned long long s = v[k];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0 && s) atomicAdd(&p.stats[k], s);
  }
  for (int k = 0; k < 16; ++k) {
    unsigned long long s = cnt.cyc[k];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0 && s) atomicAdd(&p.stats[16 + k], s);
  }
}

// COMPACT (the asm box loop's layout, see walk_to_leaf_lds): the 64-byte node is split into a 32-byte box record
// (centre.xyz half.x | half.yz | leaf word) and a 32-byte link record in a second array; the eight per-octant miss links (node
// indices) become eight words "hit code | miss code << 16": a code below 0x8000 is the LDS byte address of the next
// box (hit: the child a ray of that octant visits first — left = node + 1, or the right child when the ray runs against
// the split axis; miss: the old link), 0xFFFF ends the walk, and a leaf's hit code is 0x8000 | count << 11 | first triangle record.
// Box addresses must stay below 0x8000: 32 bytes per node, nodes first in LDS, so up to ~900 nodes — more than an
// LDS-resident scene can have (64 bytes of links + boxes and >= 48 bytes of triangles per node pair in 64 KB); the
// host checks it (ptamd_api.cpp: kCompactMaxNodes) and walks bigger trees from global memory.
template <int KIND, bool LDS_RESIDENT, bool COMPACT = false>
PT_DEV void stage_scene(const KParams& p, float4* s_mem, const float4*& s_nodes, const float4*& s_tris)
{
  if (LDS_RESIDENT) {
    if (KIND == 1) {
      stage_to_lds(s_mem, p.tris_brute, p.n_faces * 3);
      s_nodes = nullptr;
      s_tris = s_mem;
    } else {
      if (COMPACT) {
        const uint32_t base = (uint32_t)(uintptr_t)s_mem;
        // box addresses are 15-bit codes: the host only launches this layout for <= kCompactMaxNodes nodes and the
        // kernels keep no static LDS in front of s_mem; should either ever change, stop here rather than walk garbage
        // ... and a leaf's hit code holds its triangle range: count (<= 15) << 11 | first record (the host launches this layout
        // for at most kCompactMaxTris records, so that no code collides with 0xFFFF)
        if (base + p.n_nodes * 32u > 0x8000u || p.n_bvh_tris > 2047u) __builtin_trap();
        for (uint32_t i = threadIdx.x; i < p.n_nodes; i += blockDim.x) {
          const float4 q0 = p.nodes[i * 4 + 0], q1 = p.nodes[i * 4 + 1];
          const uint4 m0 = *reinterpret_cast<const uint4*>(p.nodes + i * 4 + 2), m1 = *reinterpret_cast<const uint4*>(p.nodes + i * 4 + 3);
          const uint32_t miss[8] = { m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w };
          const uint32_t child = f_as_u(q1.w), right = child & 0x3FFFFFFFu, axis = child >> 30;
          const bool leaf = (f_as_u(q0.w) >> 24) != 0u;
          uint32_t word[8];
          for (uint32_t o = 0; o < 8; ++o) {
            const uint32_t down = ((o >> axis) & 1u) ? right : i + 1u;
            // hit code: a leaf parks at itself; an interior node sends the ray to its nearer child
            const uint32_t ha = leaf ? (0x8000u | ((f_as_u(q0.w) >> 24) << 11) | (f_as_u(q0.w) & 0x7FFu)) : base + down * 32u;
            const uint32_t ma = miss[o] == PT_END ? 0xFFFFu : base + miss[o] * 32u;
            word[o] = ha | (ma << 16);
          }
          float4* links = s_mem + p.n_nodes * 2u;
          // the box as centre and half extent (the loop then needs no min / max per axis: walk_to_leaf_lds_state).  The half
          // extent is rounded up — [c - h, c + h] contains [lo, hi] —; a box that is not finite becomes "everything".
          float c[3], h[3];
          const float lo[3] = { q0.x, q0.y, q0.z }, hi[3] = { q1.x, q1.y, q1.z };
          for (int a = 0; a < 3; ++a) {
            c[a] = 0.5f * lo[a] + 0.5f * hi[a];
            h[a] = __builtin_fmaxf(hi[a] - c[a], c[a] - lo[a]) * 1.00000024f;   // (1 + 2^-22) covers the subtraction's half ulp
            if (!(__builtin_fabsf(lo[a]) <= 3.0e38f && __builtin_fabsf(hi[a]) <= 3.0e38f)) { c[a] = 0.0f; h[a] = 3.0e38f; }
          }
          s_mem[i * 2 + 0] = make_float4(c[0], c[1], c[2], h[0]);
          s_mem[i * 2 + 1] = make_float4(h[1], h[2], q0.w, 0.0f);
          links[i * 2 + 0] = make_float4(u_as_f(word[0]), u_as_f(word[1]), u_as_f(word[2]), u_as_f(word[3]));
          links[i * 2 + 1] = make_float4(u_as_f(word[4]), u_as_f(word[5]), u_as_f(word[6]), u_as_f(word[7]));
        }
      } else {
        stage_to_lds(s_mem, p.nodes, p.n_nodes * 4);
      }
      stage_to_lds(s_mem + p.n_nodes * 4, p.tris_bvh, p.n_bvh_tris * 3);
      s_nodes = s_mem;
      s_tris = s_mem + p.n_nodes * 4;
    }
    __syncthreads();
  } else {
    s_nodes = p.nodes;
    s_tris = KIND == 1 ? p.tris_brute : p.tris_bvh;
  }
}

// ---------------------------------------------------------------- the megakernel, tile form

// One thread per pixel, as in the reference.  Workgroup = 256 threads = 4 waves; wave w owns
// the 8x8 tile (w&1, w>>1) of a 16x16 block.  Lanes whose path ended idle until the wave is done.
template <int KIND, bool LDS_RESIDENT, bool STATS, int BLOCK>
__global__ void __launch_bounds__(BLOCK, PT_TILE_WAVES_PER_EU) pt_megakernel(PT_KERNEL_PARAMS)
{
  extern __shared__ float4 s_mem[];
  const float4* s_nodes;
  const float4* s_tris;
  stage_scene<KIND, LDS_RESIDENT, LDS_RESIDENT && !STATS && PT_ASM_WALK>(p, s_mem, s_nodes, s_tris);

  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  // BLOCK/64 waves as a 2 x (BLOCK/128) grid of 8x8 tiles: the block covers 16 x (BLOCK/16) pixels
  const uint32_t x = blockIdx.x * 16u + (wave & 1u) * 8u + (lane & 7u);
  const uint32_t y = p.row_begin + blockIdx.y * (BLOCK / 16u) + (wave >> 1) * 8u + (lane >> 3);
  Counters cnt = {};
  const bool active = x < p.width && y < p.row_end;
  if (active) {
    Path st;
    path_begin(p, x, y, st);
    while (!path_step<KIND, STATS, LDS_RESIDENT>(p, s_nodes, s_tris, st, cnt)) {}
    path_finish(p, st);
  }
  flush_counters<STATS>(p, cnt, active ? 1u : 0u);
}

// ---------------------------------------------------------------- the megakernel, persistent form

// Persistent waves with lane refill ("wavefront-level compaction/restart").  The grid is sized
// to the machine, the scene is staged into LDS once per workgroup, and every wave64 pulls 8x8
// pixel tiles from a global ticket counter.  A lane whose path has ended does not idle: once at
// least p.refill_min lanes are idle (ballot + popcount), they write their pixels and take the next
// pixels of the wave's tile queue (rank among idle lanes = mbcnt of the ballot), so the BVH walk
// always runs with a nearly full exec mask and lanes at different bounce depths share it.
template <int KIND, bool LDS_RESIDENT, bool STATS>
__global__ void __launch_bounds__(PT_PERSISTENT_THREADS, PT_PERSISTENT_WAVES_PER_EU) pt_megakernel_persistent(PT_KERNEL_PARAMS)
{
  extern __shared__ float4 s_mem[];
  const float4* s_nodes;
  const float4* s_tris;
  stage_scene<KIND, LDS_RESIDENT, LDS_RESIDENT && !STATS && PT_ASM_WALK>(p, s_mem, s_nodes, s_tris);

  const uint32_t lane = threadIdx.x & 63u;
  Counters cnt = {};
  uint32_t samples = 0;

  Path st;
  st.o = st.d = st.throughput = st.acc = mk3(0.f);
  st.rng.v0 = st.rng.v1 = st.rng.v2 = st.rng.v3 = st.rng.v4 = st.rng.d = 0;
  st.specular_col = 0.f; st.xy = 0; st.bk = 0;
  bool idle = true;        // this lane has no path in flight
  bool has_output = false; // ... but holds a finished, not yet written sample
  // wave-uniform tile queue.  A ticket is `tiles_per_ticket` consecutive 8x8 tiles; a wave's
  // first ticket is its own global wave index (no atomic), later ones come from the shared
  // counter, which starts at the number of waves.  One returning atomic per ~256+ pixels keeps
  // the single counter far below its ~88 dequeues/us saturation point (MI355X_MICROARCH.md).
  uint32_t tile_x0 = 0, tile_y0 = 0, tile_k = 0, qpos = 64;
  uint32_t tile = 0, tile_end = 0; // tiles [tile, tile_end) of the current ticket remain
  uint32_t ticket = blockIdx.x * (PT_PERSISTENT_THREADS / 64u) + (threadIdx.x >> 6);
  uint32_t head = blockIdx.x & 7u, dry = 0;   // workgroups are dealt round-robin to the 8 XCDs: blockIdx & 7 = this XCD's head
  bool have_ticket = true;         // `ticket` not yet expanded into tiles
  bool exhausted = false;

  for (;;) {
    const unsigned long long idle_mask = __ballot(idle);
    if ((uint32_t)__popcll(idle_mask) >= p.refill_min || idle_mask == ~0ull) {
      if (idle && has_output) {
        if (p.sample_count > 1u) path_finish_sample(p, st);
        else path_finish(p, st);
        has_output = false;
        if (STATS) samples++;
      }
      unsigned long long need = idle_mask;
      while (need) {
        if (qpos >= 64u) {
          if (exhausted) break;
          if (tile >= tile_end) {
            const uint32_t total = p.n_tiles * p.sample_count; // (tile, frame) pairs
            if (!have_ticket) {
              // pull from this workgroup's XCD head; a head that has run dry sends the wave on to the next one for good
              for (;;) {
                uint32_t t = 0;
                if (lane == 0) t = atomicAdd(p.tile_heads + head * PT_HEAD_STRIDE, 1u);
                t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
                ticket = p.n_static + t * 8u + head;
                if ((unsigned long long)ticket * p.tiles_per_ticket < total) break;
                head = (head + 1u) & 7u;
                if (++dry == 8u) break;
              }
              if (dry == 8u) { exhausted = true; break; }
            }
            have_ticket = false;
            tile = ticket * p.tiles_per_ticket;
            if (tile >= total) { exhausted = true; break; }
            tile_end = tile + p.tiles_per_ticket;
            if (tile_end > total) tile_end = total;
          }
          tile_k = tile / p.n_tiles;
          const uint32_t tl = tile - tile_k * p.n_tiles;
          tile_x0 = (tl % p.tiles_x) * PT_TILE_W;
          tile_y0 = p.row_begin + (tl / p.tiles_x) * PT_TILE_H;
          ++tile;
          qpos = 0;
        }
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
        const uint32_t avail = 64u - qpos;
        if (idle && rank < avail) {
          const uint32_t k = qpos + rank;
          const uint32_t x = tile_x0 + (k & (PT_TILE_W - 1u)), y = tile_y0 + (k >> PT_TILE_W_LOG2);
          if (x < p.width && y < p.row_end) {
            path_begin(p, x, y, st, tile_k);
            idle = false;
            has_output = true;
          }
        }
        const uint32_t wanted = (uint32_t)__popcll(need);
        qpos += wanted < avail ? wanted : avail;
        need = __ballot(idle);
      }
    }
    if (__ballot(!idle) == 0ull) break;
    if (!idle) idle = path_step<KIND, STATS, LDS_RESIDENT>(p, s_nodes, s_tris, st, cnt);
  }
  flush_counters<STATS>(p, cnt, samples);
}

// ---------------------------------------------------------------- the megakernel, restart form

// Persistent waves whose lanes are asynchronous at the level of the WALK ("wavefront-level restart").
//
// Measured on MI355X (profiles/r02_*): the persistent kernel saturates VALU issue (98 % of the issue slots at the
// clock the chip holds), with 25.6 of 64 lanes active — a wave's box-test loop runs for as long as its LONGEST walk
// (mean 15.5 nodes per ray, ~47 wave iterations per bounce), so the only way to go faster is to issue fewer
// wave-instructions per path.  Here a round of the bounce loop no longer waits for the stragglers:
//   * every lane with a path walks; the round's traversal stops as soon as fewer than `round_min` lanes (or a quarter
//     of the lanes that entered, whichever is smaller) are still unfinished;
//   * lanes whose walk completed shade (light loop + path_post) and start their next walk in the next round; the few
//     stragglers keep their place in the tree (next node + best hit so far: 5 registers) and simply continue, so the
//     box loop runs for the ~80th percentile of the walk lengths instead of their maximum;
//   * a lane whose path ended restarts at once on a fresh path.  Fresh paths come from a POOL that the wave fills 64 at
//     a time with all lanes active (seed, generateRay, camera_dof for one whole 8x8 tile: 12 dwords per path, in a
//     3 KiB slab private to the wave, L2-resident), so the expensive prologue never runs at partial occupancy.
// Every sample is parked (path_finish_sample) and pt_resolve_kernel accumulates and tonemaps, also for one frame per
// launch.  Results are bit-identical to every other variant: a path's arithmetic never depends on when it runs.
PT_DEV void pool_store(float4* slab, uint32_t e, const Path& st)
{
  // plane-major: one wave-wide store of a plane is 1 KiB contiguous
  slab[e] = make_float4(st.o.x, st.o.y, st.o.z, st.d.x);
  slab[64u + e] = make_float4(st.d.y, st.d.z, u_as_f(st.rng.v0), u_as_f(st.rng.v1));
  slab[128u + e] = make_float4(u_as_f(st.rng.v2), u_as_f(st.rng.v3), u_as_f(st.rng.v4), u_as_f(st.rng.d));
}

// The same pool in LDS, when the workgroup's share has room next to the staged scene (2 304 bytes per wave): 9 dwords
// per path, one 64-dword plane each (a wave's stores and a run of restarting lanes' loads are conflict-free).  Of the
// XORWOW state only v2, v3, v4 are kept: path_begin draws exactly twice (camera_dof), after which v0 and v1 are the
// generator's seed-independent initial v2 and v3, and the Weyl counter follows from v2 (= initial v4 = 5783321 + t0).
PT_DEV void pool_store_lds(uint32_t* pool, uint32_t e, const Path& st)
{
  pool[e] = f_as_u(st.o.x); pool[64u + e] = f_as_u(st.o.y); pool[128u + e] = f_as_u(st.o.z);
  pool[192u + e] = f_as_u(st.d.x); pool[256u + e] = f_as_u(st.d.y); pool[320u + e] = f_as_u(st.d.z);
  pool[384u + e] = st.rng.v2; pool[448u + e] = st.rng.v3; pool[512u + e] = st.rng.v4;
}

PT_DEV void pool_load_lds(const uint32_t* pool, uint32_t e, Path& st)
{
  st.o = mk3(u_as_f(pool[e]), u_as_f(pool[64u + e]), u_as_f(pool[128u + e]));
  st.d = mk3(u_as_f(pool[192u + e]), u_as_f(pool[256u + e]), u_as_f(pool[320u + e]));
  st.rng.v2 = pool[384u + e]; st.rng.v3 = pool[448u + e]; st.rng.v4 = pool[512u + e];
  const uint32_t t1 = 2591861531u * 0xf7dcefddu;          // xorwow_init: seed-independent half of the scramble
  st.rng.v0 = 521288629u + t1;
  st.rng.v1 = 88675123u ^ t1;
  st.rng.d = 6615241u + t1 + (st.rng.v2 - 5783321u) + 2u * 362437u;
}

PT_DEV void pool_load(const float4* slab, uint32_t e, Path& st)
{
  // the slab is rewritten by this wave for every tile: read past the CU's vector L1 (it is write-through and may hold
  // the previous tile's lines)
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f* q = reinterpret_cast<const v4f*>(slab);
  const v4f a = __builtin_nontemporal_load(q + e);
  const v4f b = __builtin_nontemporal_load(q + 64u + e);
  const v4f c = __builtin_nontemporal_load(q + 128u + e);
  st.o = mk3(a.x, a.y, a.z);
  st.d = mk3(a.w, b.x, b.y);
  st.rng.v0 = f_as_u(b.z); st.rng.v1 = f_as_u(b.w);
  st.rng.v2 = f_as_u(c.x); st.rng.v3 = f_as_u(c.y); st.rng.v4 = f_as_u(c.z); st.rng.d = f_as_u(c.w);
}

// One round of the nearest-hit search for the lanes that call it: walks continue from (best, node) and the round ends
// once fewer than min(round_min, entering lanes / round_div) are unfinished.  node == PT_END on return: finished.
template <bool STATS, bool NODES_IN_LDS>
PT_DEV void traverse_round(const float4* nodes, const float4* tris, uint32_t n_nodes, f3 o, f3 d, Best& best, uint32_t& node,
                           uint32_t round_min, uint32_t round_div, uint32_t round_div_m16, uint32_t walk_min, bool small_det, Counters& cnt)
{
  Walk w;
  walk_init(w, o, d, n_nodes, NODES_IN_LDS && !STATS && PT_ASM_WALK);
  w.best = best;
  w.node = node;
  const uint32_t n_start = (uint32_t)__popcll(__ballot(node != PT_END));
  uint32_t t_eff = ((n_start + round_div - 1u) * round_div_m16) >> 16;   // == (n_start + round_div - 1) / round_div (KParams::round_div_m16)
  if (t_eff > round_min) t_eff = round_min;
  if (t_eff < 1u) t_eff = 1u;
  if (STATS) w.alive = (uint32_t)__popcll(__ballot(1));
  const uint32_t lds_nodes = (uint32_t)(uintptr_t)nodes;
  if (NODES_IN_LDS && !STATS && PT_ASM_WALK) {
    uint32_t state = walk_lds_state(lds_nodes, w.node);
    const uint32_t lnk = walk_lds_link_offset(w);
    for (;;) {
      uint32_t leaf_first, leaf_count;
      walk_to_leaf_lds_state(state, lnk, w, leaf_first, leaf_count, walk_min);
      if (leaf_count != 0u) walk_leaf<STATS, true>(tris, w, leaf_first, leaf_count, cnt.tris, cnt.wave_tri_iters, small_det);
      if ((uint32_t)__popcll(__ballot(state != 0xFFFFu)) < t_eff) break;
    }
    w.node = walk_lds_node(lds_nodes, state);
  } else {
    for (;;) {
      uint32_t leaf_first, leaf_count;
      walk_to_leaf<STATS>(nodes, w, leaf_first, leaf_count, cnt.nodes, cnt.wave_node_iters, walk_min);
      if (leaf_count != 0u) walk_leaf<STATS>(tris, w, leaf_first, leaf_count, cnt.tris, cnt.wave_tri_iters, small_det);
      if ((uint32_t)__popcll(__ballot(w.node != PT_END)) < t_eff) break;
    }
  }
  best = w.best;
  node = w.node;
  if (STATS) { cnt.idle3[0] += w.idle_unstarted; cnt.idle3[1] += w.idle_finished; cnt.idle3[2] += w.idle_parked; }
}

// ---------------------------------------------------------------- four-wide walk (scenes that do not fit in LDS)

// Measured on MI355X (profiles/r02_pmc_atrium_binary_walk.json, 264 832 triangles): the binary walk from L2 spends 70 %
// of its wave-cycles waiting for memory — 50 dependent node loads per ray, each a full L1-miss round trip (252 cycles
// on average; only 75 % of them hit the XCD's 4 MB L2, the rest come from the Infinity Cache).  The cure is fewer
// round trips, not fewer bytes: a node of the wide tree (host/ptamd_internal.h: Bvh::nodes4) is ONE 128-byte line
// holding the boxes of its four children, fetched with eight independent 16-byte loads, so one round trip tests four
// boxes and leaves never cost a node load of their own (17.6 node visits per ray on the same scene).  Children that
// are hit go on a per-lane stack (LDS, [entry][lane] so a wave's pushes never conflict; entries past the LDS share
// spill to a global slab) farthest first in the node's order for the ray's octant; each entry carries its entry
// distance and is dropped on pop when the best hit is already nearer.  Exactness as for the binary walk: conservative
// boxes, (t, face index) minimum.
#define PT_NONE 0xFFFFFFFFu
#define PT_LEAF_BIT 0x80000000u

struct Stack4 {
  uint2* lds;          // this lane's column of the wave's LDS stack: entry k at lds[k * 64]
  uint2* spill;        // global continuation: entry k >= lds_entries at spill[(k - lds_entries) * 64]
  uint32_t lds_entries;
  // the first `top_n` nodes of the tree (breadth-first numbering: its top levels, which every ray crosses) staged in
  // LDS by the workgroup: a third to a half of all node visits never leave the CU
  const float4* top;
  uint32_t top_n;
};
#ifndef PT_RS4_THREADS
#define PT_RS4_THREADS 1024     /* workgroup size of the wide-walk instantiation: ONE workgroup per CU, so one copy of the LDS treelet
                                   (256 / 512 / 1024 threads: atrium 1177 / 1233 / 1278, tessellated indoor 2868 / 2966 / 3081 Msamples/s) */
#endif
#ifndef PT_TREELET_SOA
#define PT_TREELET_SOA 1        /* the LDS treelet chunk-major: 16-byte chunk q of node n at [q][n] instead of [n][q].  A wave's lanes read the
                                   SAME chunk of DIFFERENT nodes: node-major, 128-byte (64-byte) nodes put them all on two (four) of the sixteen
                                   4-bank groups — half of all LDS cycles of the atrium walk were bank conflicts (profiles/r04_base_*) —,
                                   chunk-major they spread over all sixteen by node index */
#endif
#define PT_TREELET_STRIDE 512u  /* nodes per chunk plane of a 128-byte-node treelet (64-byte nodes: twice as many); the treelet's LDS region is
                                   always 64 KB in this form, whatever the number of nodes staged */
#ifndef PT_PUSH_BRANCHFREE
#define PT_PUSH_BRANCHFREE 1    /* four-wide float walk: hit children pushed without branches when the LDS part of the stack has room */
#endif
#ifndef PT_RS4_WAVES_PER_EU
#define PT_RS4_WAVES_PER_EU 4   /* measured on the atrium (leaf records fetched in one batch): 4 / 5 / 6 waves per SIMD = 1069 / 954 / 798 Msamples/s (5 and 6 spill) */
#endif

// The LDS parts of the walk (stack column, treelet) are addressed through pointers that SAY they are LDS: with generic pointers the
// compiler folds "LDS or global" into one selected address and issues FLAT loads and stores for both (every pop a flat_load through
// the texture path, every push a ds_write of the reference plus a flat_store of the distance; the node fetch four flat loads).
// (HIP's uint2 / uint4 / float4 are classes whose assignment wants generic references: the LDS-typed accesses use the native
// vector types)
typedef uint32_t pt_u2v __attribute__((ext_vector_type(2)));
typedef uint32_t pt_u4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) pt_u2v* LdsU2;
typedef const __attribute__((address_space(3))) pt_u4v* LdsCU4;
PT_DEV LdsU2 stack4_lds(const Stack4& s) { return (LdsU2)(uintptr_t)(uint32_t)(uintptr_t)s.lds; }
PT_DEV LdsCU4 treelet_lds(const Stack4& s) { return (LdsCU4)(uintptr_t)(uint32_t)(uintptr_t)s.top; }
PT_DEV uint4 lds_u4(LdsCU4 q) { const pt_u4v v = *q; return make_uint4(v.x, v.y, v.z, v.w); }
PT_DEV float4 lds_f4(LdsCU4 q) { const pt_u4v v = *q; return make_float4(u_as_f(v.x), u_as_f(v.y), u_as_f(v.z), u_as_f(v.w)); }

PT_DEV void stack4_write(const Stack4& s, uint32_t k, uint32_t ref, float tnear)
{
  if (k < s.lds_entries) { pt_u2v e; e.x = ref; e.y = f_as_u(tnear); stack4_lds(s)[k * 64u] = e; }
  else s.spill[(size_t)(k - s.lds_entries) * 64u] = make_uint2(ref, f_as_u(tnear));
}

PT_DEV uint2 stack4_read(const Stack4& s, uint32_t k)
{
  if (k < s.lds_entries) { const pt_u2v e = stack4_lds(s)[k * 64u]; return make_uint2(e.x, e.y); }
  return s.spill[(size_t)(k - s.lds_entries) * 64u];
}

#ifndef PT_POP_BATCH
#define PT_POP_BATCH 1          /* pops after a leaf read four stack entries per LDS round trip (stack4_pop4) */
#endif
// The pop that follows a leaf.  A hit has usually just shortened the ray, so most of what the lane stacked on its way down lies
// beyond it now: the plain loop below discards those entries one LDS round trip at a time (a leaf phase of the atrium took 7 000
// cycles, most of them here: profiles/r04_notes.md).  This form reads the four entries under the top in ONE round trip and takes
// the first whose distance does not lie beyond the best hit; entries in the global continuation of the stack go one by one.
// (Used after leaves only: behind the hand-scheduled visit, whose pop already has its first entry read ahead, the same loop
// measured -2 % against the visit's own one-entry-at-a-time asm loop.)
PT_DEV uint32_t stack4_pop4(const Stack4& s, uint32_t& sp, float best_t)
{
  const LdsU2 col = stack4_lds(s);
  while (sp > 0u) {
    if (sp > s.lds_entries) {   // top of the stack is in the global continuation
      --sp;
      const uint2 e = s.spill[(size_t)(sp - s.lds_entries) * 64u];
      if (u_as_f(e.y) <= best_t) return e.x;
      continue;
    }
    // entries sp - 1 .. sp - 4 (indices below 0 read entry 0 again: never taken, the count says so)
    const uint32_t i0 = sp - 1u, i1 = sp >= 2u ? sp - 2u : 0u, i2 = sp >= 3u ? sp - 3u : 0u, i3 = sp >= 4u ? sp - 4u : 0u;
    pt_u2v e0 = col[i0 * 64u], e1 = col[i1 * 64u], e2 = col[i2 * 64u], e3 = col[i3 * 64u];
    asm volatile("" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3));   // (all four reads before the first compare)
    if (u_as_f(e0.y) <= best_t) { sp = i0; return e0.x; }
    if (sp >= 2u && u_as_f(e1.y) <= best_t) { sp = i1; return e1.x; }
    if (sp >= 3u && u_as_f(e2.y) <= best_t) { sp = i2; return e2.x; }
    if (sp >= 4u && u_as_f(e3.y) <= best_t) { sp = i3; return e3.x; }
    sp = sp >= 4u ? sp - 4u : 0u;
  }
  return PT_NONE;
}

// Next reference whose entry distance does not lie beyond the best hit (PT_NONE: the stack is empty — walk over).
PT_DEV uint32_t stack4_pop(const Stack4& s, uint32_t& sp, float best_t)
{
  while (sp > 0u) {
    --sp;
    uint2 e = stack4_read(s, sp);
    // (one 8-byte read: the compiler otherwise fetches the distance, compares, and only then fetches the reference — two
    // dependent round trips per pop)
    asm volatile("" : "+v"(e.x), "+v"(e.y));
    if (u_as_f(e.y) <= best_t) return e.x;
  }
  return PT_NONE;
}

#ifndef PT_ASM_FETCH
#define PT_ASM_FETCH 1          /* four-wide float walk: a node's eight 16-byte words fetched by fetch_node4 (below) instead of compiled loads */
#endif
// A visit's fetch: lanes whose node lies in the LDS treelet (cur < top_n; chunk-major, PT_TREELET_SOA) read it there, the others
// read their node's line from L2 — BOTH KINDS IN FLIGHT TOGETHER.  Compiled, the two branches load into the same registers and the
// compiler orders them: the global loads first, then `s_waitcnt vmcnt(..)` in front of every ds_read (it cannot know that the two
// write disjoint lanes), so the treelet's LDS latency — bank conflicts included — came AFTER the memory latency in every box
// iteration of the wave (profiles/r04_notes.md).  Loads only write the lanes EXEC enables, so no order is needed.
typedef float pt_f4v __attribute__((ext_vector_type(4)));
template <uint32_t PLANE_BYTES = PT_TREELET_STRIDE * 16u>   /* bytes between the chunk planes of the treelet */
PT_DEV void fetch_node4(const float4* nodes4, const Stack4& stk, uint32_t cur, uint32_t sp, pt_f4v& r0, pt_f4v& r1, pt_f4v& r2, pt_f4v& r3, pt_f4v& r4,
                        pt_f4v& r5, pt_f4v& r6, pt_f4v& r7, pt_u2v& ahead)
{
  static_assert(PLANE_BYTES * 7u < 65536u, "chunk plane offsets must fit a ds_read offset field");
  const unsigned long long ga = (unsigned long long)(uintptr_t)nodes4 + ((unsigned long long)cur << 7);
  const uint32_t la = (uint32_t)(uintptr_t)stk.top + (cur << 4);
  const uint32_t pa = (uint32_t)(uintptr_t)stk.lds + ((sp - 1u) << 9);   // (sp == 0: an address below the column — whatever is there is never used)
  unsigned long long save;
  asm volatile(
      "ds_read_b64 %[ah], %[pa]\n\t"
      "v_cmp_le_u32 vcc, %[topn], %[cur]\n\t"
      "s_and_saveexec_b64 %[save], vcc\n\t"                 // EXEC = lanes whose node is NOT in the treelet: the long latency first
      "global_load_dwordx4 %[r0], %[ga], off\n\t"
      "global_load_dwordx4 %[r1], %[ga], off offset:16\n\t"
      "global_load_dwordx4 %[r2], %[ga], off offset:32\n\t"
      "global_load_dwordx4 %[r3], %[ga], off offset:48\n\t"
      "global_load_dwordx4 %[r4], %[ga], off offset:64\n\t"
      "global_load_dwordx4 %[r5], %[ga], off offset:80\n\t"
      "global_load_dwordx4 %[r6], %[ga], off offset:96\n\t"
      "global_load_dwordx4 %[r7], %[ga], off offset:112\n\t"
      "s_andn2_b64 exec, %[save], exec\n\t"                 // EXEC = the treelet's lanes of the visit
      "ds_read_b128 %[r0], %[la]\n\t"
      "ds_read_b128 %[r1], %[la] offset:%[p1]\n\t"
      "ds_read_b128 %[r2], %[la] offset:%[p2]\n\t"
      "ds_read_b128 %[r3], %[la] offset:%[p3]\n\t"
      "ds_read_b128 %[r4], %[la] offset:%[p4]\n\t"
      "ds_read_b128 %[r5], %[la] offset:%[p5]\n\t"
      "ds_read_b128 %[r6], %[la] offset:%[p6]\n\t"
      "ds_read_b128 %[r7], %[la] offset:%[p7]\n\t"
      "s_mov_b64 exec, %[save]\n\t"
      "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
      : [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3), [r4] "=&v"(r4), [r5] "=&v"(r5), [r6] "=&v"(r6), [r7] "=&v"(r7),
        [ah] "=&v"(ahead), [save] "=&s"(save)
      : [topn] "s"(stk.top_n), [cur] "v"(cur), [la] "v"(la), [ga] "v"(ga), [pa] "v"(pa),
        [p1] "i"(PLANE_BYTES), [p2] "i"(2u * PLANE_BYTES), [p3] "i"(3u * PLANE_BYTES), [p4] "i"(4u * PLANE_BYTES), [p5] "i"(5u * PLANE_BYTES),
        [p6] "i"(6u * PLANE_BYTES), [p7] "i"(7u * PLANE_BYTES)
      : "vcc", "scc", "memory");
}
#ifndef PT_ASM_SELECT
#define PT_ASM_SELECT 1         /* four-wide float walk: what follows the box tests of a visit — ranks, pushes, the nearest child or a pop — hand-scheduled */
#endif
// Second half of a visit of the four-wide float walk, hand-scheduled: from the four box-test results (m0..m3: the lanes whose
// child c was hit; tn: entry distances; refs / ord: the node's references and order words) to the reference the lane processes
// next.  Hit children that are not the nearest go on the lane's LDS stack at sp + rank (rank = hit children farther along the
// ray, read off the node's order word for the ray's octant); every other child is written to the free slot sp + 3, so that the four
// stores need no branch; the nearest hit child is returned; a visit without a hit pops — first the entry read ahead with the node
// (`ahead`), then further ones through LDS — until an entry's distance does not lie beyond the best hit (PT_NONE: stack empty).
// Same semantics as walk4_compute's tail (which remains the path of waves whose LDS stack part is nearly full); compiled, this
// part of the visit was ~75 VALU and ~60 SALU instructions with a dozen branches, here 48 VALU, 14 SALU and one branch, and the
// LDS round trip of a pop is off the critical path.  Scratch: v64..v77 (clobbered; wide kernels have a budget of 128 VGPRs).
PT_DEV uint32_t walk4_select_asm(const Stack4& stk, const Walk& w, unsigned long long m0, unsigned long long m1, unsigned long long m2,
                                 unsigned long long m3, float tn0, float tn1, float tn2, float tn3, pt_f4v refs, pt_f4v ord, pt_u2v ahead, uint32_t& sp)
{
  uint32_t cur;
  unsigned long long save;
  const uint32_t col = (uint32_t)(uintptr_t)stk.lds;
  asm volatile(
      "s_nop 1\n\t"                                          // m0..m3 were written by VALU compares
      "v_cndmask_b32_e64 v64, 0, 1, %[m0]\n\t"
      "v_cndmask_b32_e64 v65, 0, 2, %[m1]\n\t"
      "v_cndmask_b32_e64 v66, 0, 4, %[m2]\n\t"
      "v_cndmask_b32_e64 v67, 0, 8, %[m3]\n\t"
      "v_or3_b32 v64, v64, v65, v66\n\t"
      "v_or_b32 v64, v64, v67\n\t"                           // v64: hit bits
      // the order word of the ray's octant: halfword `oct` of ord.xyzw (nibble c: the children visited AFTER child c)
      "v_cndmask_b32_e64 v65, %[o0], %[o1], %[oy]\n\t"
      "v_cndmask_b32_e64 v66, %[o2], %[o3], %[oy]\n\t"
      "v_cndmask_b32_e64 v65, v65, v66, %[oz]\n\t"
      "v_lshrrev_b32 v66, 16, v65\n\t"
      "v_cndmask_b32_e64 v65, v65, v66, %[ox]\n\t"           // (bits above the halfword are masked off by the AND with the four hit bits)
      "v_bcnt_u32_b32 v67, v64, 0\n\t"                       // v67: hit children
      "v_add_u32 v68, -1, v67\n\t"                           // v68: the nearest one's rank (-1: none)
      "v_and_b32 v69, v64, v65\n\t"
      "v_bcnt_u32_b32 v69, v69, 0\n\t"                       // v69..v72: rank of child 0..3 = hit children farther than it
      "v_lshrrev_b32 v70, 4, v65\n\t"
      "v_and_b32 v70, v64, v70\n\t"
      "v_bcnt_u32_b32 v70, v70, 0\n\t"
      "v_lshrrev_b32 v71, 8, v65\n\t"
      "v_and_b32 v71, v64, v71\n\t"
      "v_bcnt_u32_b32 v71, v71, 0\n\t"
      "v_lshrrev_b32 v72, 12, v65\n\t"
      "v_and_b32 v72, v64, v72\n\t"
      "v_bcnt_u32_b32 v72, v72, 0\n\t"
      "v_lshl_add_u32 v73, %[sp], 9, %[col]\n\t"              // v73: LDS address of entry sp of this lane's column
      "v_mov_b32 %[cur], -1\n\t"
#define PT_SELECT_CHILD(RK, M, F, T)                                                                       \
      "v_cmp_eq_u32 vcc, " RK ", v68\n\t"                                                                   \
      "s_and_b64 vcc, vcc, " M "\n\t"                        /* the nearest hit child: visited next */       \
      "v_cndmask_b32 %[cur], %[cur], " F ", vcc\n\t"                                                        \
      "s_andn2_b64 vcc, " M ", vcc\n\t"                      /* hit, not the nearest: stacked at sp + rank */ \
      "v_cndmask_b32 " RK ", 3, " RK ", vcc\n\t"                                                           \
      "v_lshl_add_u32 " RK ", " RK ", 9, v73\n\t"                                                           \
      "ds_write2_b32 " RK ", " F ", " T " offset1:1\n\t"
      PT_SELECT_CHILD("v69", "%[m0]", "%[f0]", "%[t0]")
      PT_SELECT_CHILD("v70", "%[m1]", "%[f1]", "%[t1]")
      PT_SELECT_CHILD("v71", "%[m2]", "%[f2]", "%[t2]")
      PT_SELECT_CHILD("v72", "%[m3]", "%[f3]", "%[t3]")
#undef PT_SELECT_CHILD
      "v_add_u32 %[sp], %[sp], v68\n\t"                       // sp += hit children - 1 (no hit: the entry below is the candidate)
      // ---- lanes whose visit hit nothing pop
      "v_cmp_eq_u32 vcc, 0, v67\n\t"
      "s_and_saveexec_b64 %[save], vcc\n\t"
      "s_cbranch_execz 3f\n\t"
      "v_cmp_lt_i32 vcc, -1, %[sp]\n\t"                       // an entry is left (else the stack is empty: PT_NONE)
      "s_and_b64 exec, exec, vcc\n\t"
      "s_cbranch_execz 2f\n\t"
      "v_cmp_le_f32 vcc, %[a1], %[best]\n\t"                  // the entry read ahead: taken unless its distance lies beyond the best hit
      "s_nop 1\n\t"
      "v_cndmask_b32 %[cur], %[cur], %[a0], vcc\n\t"
      "s_andn2_b64 exec, exec, vcc\n\t"
      "s_cbranch_execz 2f\n\t"
      "1:\n\t"
      "v_add_u32 %[sp], -1, %[sp]\n\t"
      "v_cmp_lt_i32 vcc, -1, %[sp]\n\t"
      "s_and_b64 exec, exec, vcc\n\t"
      "s_cbranch_execz 2f\n\t"
      "v_lshl_add_u32 v69, %[sp], 9, %[col]\n\t"
      "ds_read_b64 v[76:77], v69\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_cmp_le_f32 vcc, v77, %[best]\n\t"
      "s_nop 1\n\t"
      "v_cndmask_b32 %[cur], %[cur], v76, vcc\n\t"
      "s_andn2_b64 exec, exec, vcc\n\t"
      "s_cbranch_execnz 1b\n\t"
      "2:\n\t"
      "3:\n\t"
      "s_mov_b64 exec, %[save]\n\t"
      "v_max_i32 %[sp], %[sp], 0\n\t"                         // (a lane that ran out of entries ended at -1)
      : [cur] "=&v"(cur), [sp] "+v"(sp), [save] "=&s"(save)
      : [m0] "s"(m0), [m1] "s"(m1), [m2] "s"(m2), [m3] "s"(m3), [ox] "s"(w.oct_x), [oy] "s"(w.oct_y), [oz] "s"(w.oct_z),
        [t0] "v"(tn0), [t1] "v"(tn1), [t2] "v"(tn2), [t3] "v"(tn3), [f0] "v"(refs.x), [f1] "v"(refs.y), [f2] "v"(refs.z), [f3] "v"(refs.w),
        [o0] "v"(ord.x), [o1] "v"(ord.y), [o2] "v"(ord.z), [o3] "v"(ord.w), [a0] "v"(ahead.x), [a1] "v"(ahead.y), [best] "v"(w.best.t), [col] "v"(col)
      : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v76", "v77", "vcc", "scc", "memory");
  return cur;
}

// The arithmetic of a visit on a fetched node: tests the four child boxes, stacks the hit ones, returns the nearest (or pops).
PT_DEV uint32_t walk4_compute(float4 lx, float4 ly, float4 lz, float4 hx, float4 hy, float4 hz, uint4 refs, uint4 ord, const Stack4& stk, const Walk& w, uint32_t& sp);

// Visits interior node `cur`: fetches it (LDS treelet or L2), then walk4_compute.
template <bool STATS, uint32_t PLANE_BYTES = PT_TREELET_STRIDE * 16u>
PT_DEV uint32_t walk4_visit(const float4* nodes4, const Stack4& stk, const Walk& w, uint32_t cur, uint32_t& sp, Counters* cnt = nullptr)
{
  float4 lx, ly, lz, hx, hy, hz;
  uint4 refs, ord;
#if PT_ASM_FETCH && PT_TREELET_SOA
  {
    pt_f4v r0, r1, r2, r3, r4, r5, r6, r7;
    long long tv0 = 0;
    if (STATS) tv0 = clock64();
    pt_u2v ahead;
    fetch_node4<PLANE_BYTES>(nodes4, stk, cur, sp, r0, r1, r2, r3, r4, r5, r6, r7, ahead);
    if (STATS && cnt) { const long long tv1 = clock64(); if (PT_WAVE_ONE()) cnt->cyc[6] += (unsigned long long)(tv1 - tv0); }
#if PT_ASM_SELECT
    // Usual case, decided for the whole wave: four more entries fit into the LDS part of every lane's stack (then every entry a pop
    // of this visit can reach is in LDS too)
    if (__ballot(sp + 4u > stk.lds_entries) == 0ull) {
      const float aix = __builtin_fabsf(w.inv.x), aiy = __builtin_fabsf(w.inv.y), aiz = __builtin_fabsf(w.inv.z);
      float tn[4];
      unsigned long long m[4];
#define PT_CHILD(c, X)                                                                                                   \
      {                                                                                                                  \
        const float tcx = __builtin_fmaf(r0.X, w.inv.x, w.noi.x), tcy = __builtin_fmaf(r1.X, w.inv.y, w.noi.y),           \
                    tcz = __builtin_fmaf(r2.X, w.inv.z, w.noi.z);                                                        \
        const float n0x = __builtin_fmaf(-r3.X, aix, tcx), f0x = __builtin_fmaf(r3.X, aix, tcx);                          \
        const float n0y = __builtin_fmaf(-r4.X, aiy, tcy), f0y = __builtin_fmaf(r4.X, aiy, tcy);                          \
        const float n0z = __builtin_fmaf(-r5.X, aiz, tcz), f0z = __builtin_fmaf(r5.X, aiz, tcz);                          \
        const float tnear = __builtin_fmaxf(__builtin_fmaxf(n0x, n0y), n0z);                                             \
        const float tfar = __builtin_fminf(__builtin_fminf(f0x, f0y), f0z);                                              \
        tn[c] = __builtin_fmaxf(tnear, 0.0f);                                                                            \
        m[c] = __ballot(tn[c] <= __builtin_fminf(tfar, w.best.t));                                                       \
      }
      PT_CHILD(0, x) PT_CHILD(1, y) PT_CHILD(2, z) PT_CHILD(3, w)
#undef PT_CHILD
      return walk4_select_asm(stk, w, m[0], m[1], m[2], m[3], tn[0], tn[1], tn[2], tn[3], r6, r7, ahead, sp);
    }
#endif
    lx = make_float4(r0.x, r0.y, r0.z, r0.w); ly = make_float4(r1.x, r1.y, r1.z, r1.w); lz = make_float4(r2.x, r2.y, r2.z, r2.w);
    hx = make_float4(r3.x, r3.y, r3.z, r3.w); hy = make_float4(r4.x, r4.y, r4.z, r4.w); hz = make_float4(r5.x, r5.y, r5.z, r5.w);
    refs = make_uint4(f_as_u(r6.x), f_as_u(r6.y), f_as_u(r6.z), f_as_u(r6.w));
    ord = make_uint4(f_as_u(r7.x), f_as_u(r7.y), f_as_u(r7.z), f_as_u(r7.w));
    return walk4_compute(lx, ly, lz, hx, hy, hz, refs, ord, stk, w, sp);
  }
#endif
  if (cur < stk.top_n) {
#if PT_TREELET_SOA
    constexpr uint32_t S = PT_TREELET_STRIDE;
    const LdsCU4 q = treelet_lds(stk) + cur;
#else
    constexpr uint32_t S = 1u;
    const LdsCU4 q = treelet_lds(stk) + cur * 8u;
#endif
    lx = lds_f4(q); ly = lds_f4(q + S); lz = lds_f4(q + 2u * S); hx = lds_f4(q + 3u * S); hy = lds_f4(q + 4u * S); hz = lds_f4(q + 5u * S);
    refs = lds_u4(q + 6u * S);
    ord = lds_u4(q + 7u * S);
  } else {
    const float4* q = nodes4 + (size_t)cur * 8u;
    lx = q[0]; ly = q[1]; lz = q[2]; hx = q[3]; hy = q[4]; hz = q[5];
    refs = *reinterpret_cast<const uint4*>(q + 6);
    ord = *reinterpret_cast<const uint4*>(q + 7);
  }
  // (references and order words are needed last, but fetched with the boxes: one round trip per visit)
  asm volatile("" : "+v"(refs.x), "+v"(refs.y), "+v"(refs.z), "+v"(refs.w), "+v"(ord.x), "+v"(ord.y), "+v"(ord.z), "+v"(ord.w));
  return walk4_compute(lx, ly, lz, hx, hy, hz, refs, ord, stk, w, sp);
}

PT_DEV uint32_t walk4_compute(float4 lx, float4 ly, float4 lz, float4 hx, float4 hy, float4 hz, uint4 refs, uint4 ord, const Stack4& stk, const Walk& w, uint32_t& sp)
{
  float tn[4];
  uint32_t hit = 0;
#define PT_CHILD(c, X)                                                                                                   \
  {                                                                                                                      \
    /* (lx .. lz hold the child boxes' centres, hx .. hz their half extents: entry / exit = t(centre) -+ half * |1/d|) */  \
    const float tcx = __builtin_fmaf(lx.X, w.inv.x, w.noi.x), tcy = __builtin_fmaf(ly.X, w.inv.y, w.noi.y),               \
                tcz = __builtin_fmaf(lz.X, w.inv.z, w.noi.z);                                                            \
    const float n0x = __builtin_fmaf(-hx.X, aix, tcx), f0x = __builtin_fmaf(hx.X, aix, tcx);                              \
    const float n0y = __builtin_fmaf(-hy.X, aiy, tcy), f0y = __builtin_fmaf(hy.X, aiy, tcy);                              \
    const float n0z = __builtin_fmaf(-hz.X, aiz, tcz), f0z = __builtin_fmaf(hz.X, aiz, tcz);                              \
    const float tnear = __builtin_fmaxf(__builtin_fmaxf(n0x, n0y), n0z);                                                 \
    const float tfar = __builtin_fminf(__builtin_fminf(f0x, f0y), f0z);                                                  \
    tn[c] = __builtin_fmaxf(tnear, 0.0f);                                                                                \
    if (tn[c] <= __builtin_fminf(tfar, w.best.t)) hit |= 1u << c;                                                        \
  }
  const float aix = __builtin_fabsf(w.inv.x), aiy = __builtin_fabsf(w.inv.y), aiz = __builtin_fabsf(w.inv.z);
  PT_CHILD(0, x) PT_CHILD(1, y) PT_CHILD(2, z) PT_CHILD(3, w)
#undef PT_CHILD
  if (hit == 0u) return stack4_pop(stk, sp, w.best.t);
  // halfword `oct` of q7: nibble c = the children this octant visits after child c
  const uint32_t pair = (w.oct & 4u) ? ((w.oct & 2u) ? ord.w : ord.z) : ((w.oct & 2u) ? ord.y : ord.x);
  const uint32_t order = (w.oct & 1u) ? (pair >> 16) : (pair & 0xFFFFu);
  const uint32_t nhit = (uint32_t)__builtin_popcount(hit);
  const uint32_t ref[4] = { refs.x, refs.y, refs.z, refs.w };
  uint32_t next = PT_NONE;
#if PT_PUSH_BRANCHFREE
  // Usual case, decided for the whole wave: four more entries fit into the LDS part of every lane's stack.  Then every child is
  // written without a branch — a hit child that is not the nearest to slot sp + rank, the others to slot sp + 3, which lies at or
  // above the new top of the stack (at most three are pushed) and is free.
  if (__ballot(sp + 4u > stk.lds_entries) == 0ull) {
    const LdsU2 col = stack4_lds(stk);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bool h = ((hit >> c) & 1u) != 0u;
      const uint32_t rank = (uint32_t)__builtin_popcount(hit & (order >> (4 * c)) & 0xFu);
      const bool nearest = h && rank + 1u == nhit;
      next = nearest ? ref[c] : next;
      pt_u2v e;
      e.x = ref[c]; e.y = f_as_u(tn[c]);
      col[(sp + ((h && !nearest) ? rank : 3u)) * 64u] = e;
    }
    sp += nhit - 1u;
    return next;
  }
#endif
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if ((hit >> c) & 1u) {
      // rank = hit children farther than c: the farthest goes deepest, the nearest is visited next without a round trip
      // through the stack
      const uint32_t rank = (uint32_t)__builtin_popcount(hit & (order >> (4 * c)) & 0xFu);
      if (rank + 1u == nhit) next = ref[c];
      else stack4_write(stk, sp + rank, ref[c], tn[c]);
    }
  }
  sp += nhit - 1u;
  return next;
}

// ---------------------------------------------------------------- eight-wide walk over quantised nodes (round 3)
//
// The four-wide walk of round 2 ran at 72 % of the request rate the L2 / Infinity-Cache path sustains for random 128-byte
// records (profiles/r02_gather_ubench.txt), so the lever is requests per ray.  A node of Bvh::nodes8 (host/ptamd_internal.h)
// still is ONE 128-byte line but holds EIGHT child boxes: a float origin, one power-of-two scale per axis and 8-bit planes,
// rounded outward.  15.4 -> 9.3 node visits per ray on the atrium, and the 64 KB of LDS that held the top 512 four-wide nodes
// (2 048 boxes) now hold 4 096 boxes: two thirds of all visits stay in the CU.  Per node and axis the ray's slab coefficients
// are folded into the decode:  t = plane * (scale / d) + (origin - o) / d  — one v_cvt_f32_ubyte and one fma per plane.
// The ray's octant comes from the SIGN BITS of its direction (so that -0.0 agrees with the sign of its stand-in reciprocal);
// it selects the entry / exit plane arrays (no min / max per axis) and the visiting order: children sit in slots by direction
// (bvh_builder.cpp), ascending (slot ^ octant) is front to back.  far[c] = the slots a ray of this octant reaches AFTER slot c
// (one byte per slot, two dwords per ray, from KParams::far_table): a hit child's stack position is one popcount.

struct Walk8 {
  uint32_t oct;   // bit a: the direction's sign bit along axis a
  uint2 far;      // byte c: slots visited after slot c
};

PT_DEV Walk8 walk8_init(const KParams& p, f3 d)
{
  Walk8 x;
  x.oct = (f_as_u(d.x) >> 31) | ((f_as_u(d.y) >> 31) << 1) | ((f_as_u(d.z) >> 31) << 2);
  // per-lane read of the 64-byte table in the kernel-argument segment (one cached load per walk)
  typedef const __attribute__((address_space(4))) uint32_t* ConstU;
  const ConstU tbl = (ConstU)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + __builtin_offsetof(KParams, far_table));
  x.far = make_uint2(tbl[2u * x.oct], tbl[2u * x.oct + 1u]);
  return x;
}

// Visits interior node `cur`: tests its eight child boxes, stacks the hit ones, returns the nearest (or pops).
template <bool STATS>
PT_DEV uint32_t walk8_visit(const float4* nodes8, const Stack4& stk, const Walk& w, const Walk8& x, uint32_t cur, uint32_t& sp)
{
  uint4 hd, ra, rb, pa, pb, pc;
  if (cur < stk.top_n) {
#if PT_TREELET_SOA
    constexpr uint32_t S = PT_TREELET_STRIDE;
    const LdsCU4 q = treelet_lds(stk) + cur;
#else
    constexpr uint32_t S = 1u;
    const LdsCU4 q = treelet_lds(stk) + cur * 8u;
#endif
    hd = lds_u4(q); ra = lds_u4(q + S); rb = lds_u4(q + 2u * S); pa = lds_u4(q + 3u * S); pb = lds_u4(q + 4u * S); pc = lds_u4(q + 5u * S);
  } else {
    const uint4* q = reinterpret_cast<const uint4*>(nodes8) + (size_t)cur * 8u;
    hd = q[0]; ra = q[1]; rb = q[2]; pa = q[3]; pb = q[4]; pc = q[5];
  }
  // (the references are needed last, but fetched now: the compiler would otherwise load them after the box tests — a second
  // round trip per visit)
  asm volatile("" : "+v"(ra.x), "+v"(ra.y), "+v"(ra.z), "+v"(ra.w), "+v"(rb.x), "+v"(rb.y), "+v"(rb.z), "+v"(rb.w));
  // planes: pa = lo.x[8] lo.y[8], pb = lo.z[8] hi.x[8], pc = hi.y[8] hi.z[8]; the ray enters through `en`, leaves through `ex`
  const bool nx = (x.oct & 1u) != 0u, ny = (x.oct & 2u) != 0u, nz = (x.oct & 4u) != 0u;
  const uint32_t enx0 = nx ? pb.z : pa.x, enx1 = nx ? pb.w : pa.y, exx0 = nx ? pa.x : pb.z, exx1 = nx ? pa.y : pb.w;
  const uint32_t eny0 = ny ? pc.x : pa.z, eny1 = ny ? pc.y : pa.w, exy0 = ny ? pa.z : pc.x, exy1 = ny ? pa.w : pc.y;
  const uint32_t enz0 = nz ? pc.z : pb.x, enz1 = nz ? pc.w : pb.y, exz0 = nz ? pb.x : pc.z, exz1 = nz ? pb.y : pc.w;
  const float ax = u_as_f((hd.w & 0xFFu) << 23) * w.inv.x, ay = u_as_f(((hd.w >> 8) & 0xFFu) << 23) * w.inv.y,
              az = u_as_f(((hd.w >> 16) & 0xFFu) << 23) * w.inv.z;
  const float bx = __builtin_fmaf(u_as_f(hd.x), w.inv.x, w.noi.x), by = __builtin_fmaf(u_as_f(hd.y), w.inv.y, w.noi.y),
              bz = __builtin_fmaf(u_as_f(hd.z), w.inv.z, w.noi.z);
  float tn[8];
  uint32_t hit = 0;
#define PT_CHILD8(c, W0, SH)                                                                                              \
  {                                                                                                                      \
    const float tnx = __builtin_fmaf((float)((enx##W0 >> SH) & 0xFFu), ax, bx), tfx = __builtin_fmaf((float)((exx##W0 >> SH) & 0xFFu), ax, bx); \
    const float tny = __builtin_fmaf((float)((eny##W0 >> SH) & 0xFFu), ay, by), tfy = __builtin_fmaf((float)((exy##W0 >> SH) & 0xFFu), ay, by); \
    const float tnz = __builtin_fmaf((float)((enz##W0 >> SH) & 0xFFu), az, bz), tfz = __builtin_fmaf((float)((exz##W0 >> SH) & 0xFFu), az, bz); \
    tn[c] = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(tnx, tny), tnz), 0.0f);                                      \
    const float tf = __builtin_fminf(__builtin_fminf(__builtin_fminf(tfx, tfy), tfz), w.best.t);                         \
    if (tn[c] <= tf) hit |= 1u << c;                                                                                     \
  }
  PT_CHILD8(0, 0, 0) PT_CHILD8(1, 0, 8) PT_CHILD8(2, 0, 16) PT_CHILD8(3, 0, 24)
  PT_CHILD8(4, 1, 0) PT_CHILD8(5, 1, 8) PT_CHILD8(6, 1, 16) PT_CHILD8(7, 1, 24)
#undef PT_CHILD8
  if (hit == 0u) return stack4_pop(stk, sp, w.best.t);
  const uint32_t nhit = (uint32_t)__builtin_popcount(hit);
  const uint32_t ref[8] = { ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w };
  uint32_t next = PT_NONE;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    if ((hit >> c) & 1u) {
      // hit children farther than c go deeper; the nearest is visited next without a round trip through the stack
      const uint32_t farther = ((c < 4 ? x.far.x : x.far.y) >> (8 * (c & 3))) & 0xFFu;
      const uint32_t rank = (uint32_t)__builtin_popcount(hit & farther);
      if (rank + 1u == nhit) next = ref[c];
      else stack4_write(stk, sp + rank, ref[c], tn[c]);
    }
  }
  sp += nhit - 1u;
  return next;
}

// ---------------------------------------------------------------- four-wide walk over 64-byte quantised nodes (late round 3)
//
// The same tree and numbering as the float form (Bvh::nodes4q): origin, per-axis power-of-two scale, 8-bit child planes rounded
// outward, references, and the visiting order of octants 0..3 (an octant's opposite visits in the reverse order).  A visit is
// FOUR 16-byte loads instead of eight: the CU's memory pipe is what binds this walk (doubling the loads of a visit, all of them L1
// hits, cost 40-68 %: HISTORY.md).  The planes a ray enters / leaves through follow the SIGN BIT of its direction, as in the
// eight-wide form.
template <bool STATS>
PT_DEV uint32_t walk4q_visit(const float4* nodesq, const Stack4& stk, const Walk& w, uint32_t soct, uint32_t cur, uint32_t& sp)
{
  uint4 hd, rf, pa, pb;
  if (cur < stk.top_n) {
#if PT_TREELET_SOA
    constexpr uint32_t S = 2u * PT_TREELET_STRIDE;
    const LdsCU4 q = treelet_lds(stk) + cur;
#else
    constexpr uint32_t S = 1u;
    const LdsCU4 q = treelet_lds(stk) + cur * 4u;
#endif
    hd = lds_u4(q); rf = lds_u4(q + S); pa = lds_u4(q + 2u * S); pb = lds_u4(q + 3u * S);
  } else {
    const uint4* q = reinterpret_cast<const uint4*>(nodesq) + (size_t)cur * 4u;
    hd = q[0]; rf = q[1]; pa = q[2]; pb = q[3];
  }
  // (references and order words are needed last, but fetched with the planes: one round trip per visit)
  asm volatile("" : "+v"(rf.x), "+v"(rf.y), "+v"(rf.z), "+v"(rf.w), "+v"(pb.z), "+v"(pb.w));
  // planes: pa = lo.x[4] lo.y[4] lo.z[4] hi.x[4], pb.xy = hi.y[4] hi.z[4]
  const bool nx = (soct & 1u) != 0u, ny = (soct & 2u) != 0u, nz = (soct & 4u) != 0u;
  const uint32_t enx = nx ? pa.w : pa.x, exx = nx ? pa.x : pa.w;
  const uint32_t eny = ny ? pb.x : pa.y, exy = ny ? pa.y : pb.x;
  const uint32_t enz = nz ? pb.y : pa.z, exz = nz ? pa.z : pb.y;
  const float ax = u_as_f((hd.w & 0xFFu) << 23) * w.inv.x, ay = u_as_f(((hd.w >> 8) & 0xFFu) << 23) * w.inv.y,
              az = u_as_f(((hd.w >> 16) & 0xFFu) << 23) * w.inv.z;
  const float bx = __builtin_fmaf(u_as_f(hd.x), w.inv.x, w.noi.x), by = __builtin_fmaf(u_as_f(hd.y), w.inv.y, w.noi.y),
              bz = __builtin_fmaf(u_as_f(hd.z), w.inv.z, w.noi.z);
  float tn[4];
  uint32_t hit = 0;
#define PT_CHILDQ(c, SH)                                                                                                 \
  {                                                                                                                      \
    const float tnx = __builtin_fmaf((float)((enx >> SH) & 0xFFu), ax, bx), tfx = __builtin_fmaf((float)((exx >> SH) & 0xFFu), ax, bx); \
    const float tny = __builtin_fmaf((float)((eny >> SH) & 0xFFu), ay, by), tfy = __builtin_fmaf((float)((exy >> SH) & 0xFFu), ay, by); \
    const float tnz = __builtin_fmaf((float)((enz >> SH) & 0xFFu), az, bz), tfz = __builtin_fmaf((float)((exz >> SH) & 0xFFu), az, bz); \
    tn[c] = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(tnx, tny), tnz), 0.0f);                                      \
    const float tf = __builtin_fminf(__builtin_fminf(__builtin_fminf(tfx, tfy), tfz), w.best.t);                         \
    if (tn[c] <= tf) hit |= 1u << c;                                                                                     \
  }
  PT_CHILDQ(0, 0) PT_CHILDQ(1, 8) PT_CHILDQ(2, 16) PT_CHILDQ(3, 24)
#undef PT_CHILDQ
  if (hit == 0u) return stack4_pop(stk, sp, w.best.t);
  // halfword h of pb.zw: the order of octant h (0..3); octants 4..7 read the opposite octant's, inverted — every nibble then also
  // names its own child, which the rank below takes off again
  const uint32_t h = nz ? (~soct & 3u) : soct;
  uint32_t order = ((h & 2u) ? pb.w : pb.z) >> ((h & 1u) << 4);
  if (nz) order = ~order;
  const uint32_t self_counted = nz ? 1u : 0u;
  const uint32_t nhit = (uint32_t)__builtin_popcount(hit);
  const uint32_t ref[4] = { rf.x, rf.y, rf.z, rf.w };
  uint32_t next = PT_NONE;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if ((hit >> c) & 1u) {
      const uint32_t rank = (uint32_t)__builtin_popcount(hit & (order >> (4 * c)) & 0xFu) - self_counted;
      if (rank + 1u == nhit) next = ref[c];
      else stack4_write(stk, sp + rank, ref[c], tn[c]);
    }
  }
  sp += nhit - 1u;
  return next;
}

// One round of the wide walk for the lanes that call it (see traverse_round): cur == PT_NONE on return: finished.
// MODE: 0 four-wide float nodes, 1 eight-wide quantised nodes, 2 four-wide quantised nodes
template <bool STATS, int MODE = 0>
PT_DEV void traverse_round4(const KParams& p, const float4* nodes4, const float4* tris, const Stack4& stk, f3 o, f3 d, Best& best, uint32_t& cur,
                            uint32_t& sp, uint32_t round_min, uint32_t round_div, uint32_t round_div_m16, uint32_t walk_min, bool small_det, Counters& cnt)
{
  Walk w;
  walk_init(w, o, d, 1u);
  w.best = best;
  w.oct_x = __ballot((w.oct & 1u) != 0u); w.oct_y = __ballot((w.oct & 2u) != 0u); w.oct_z = __ballot((w.oct & 4u) != 0u);
  Walk8 x8;
  x8.oct = 0u; x8.far = make_uint2(0u, 0u);
  if (MODE == 1) x8 = walk8_init(p, d);
  if (MODE == 2) x8.oct = (f_as_u(d.x) >> 31) | ((f_as_u(d.y) >> 31) << 1) | ((f_as_u(d.z) >> 31) << 2);
  const uint32_t n_start = (uint32_t)__popcll(__ballot(cur != PT_NONE));
  uint32_t t_eff = ((n_start + round_div - 1u) * round_div_m16) >> 16;   // == (n_start + round_div - 1) / round_div (KParams::round_div_m16)
  if (t_eff > round_min) t_eff = round_min;
  if (t_eff < 1u) t_eff = 1u;
  for (;;) {
    long long tb0 = 0;
    if (STATS) tb0 = clock64();
    // box phase: every lane that holds an interior node visits it; the phase ends when fewer than walk_min lanes do
    for (;;) {
      const bool interior = cur < PT_LEAF_BIT;
      const uint32_t walkers = (uint32_t)__popcll(__ballot(interior));
      if (walkers == 0u) break;
      if (STATS) {   // (every lane of the call is here) idle lane-slots of this box iteration, by cause
        const uint32_t n_call = (uint32_t)__popcll(__ballot(1)), n_over = (uint32_t)__popcll(__ballot(cur == PT_NONE)),
                       n_parked = (uint32_t)__popcll(__ballot(cur != PT_NONE && cur >= PT_LEAF_BIT));
        if (PT_WAVE_ONE()) {
          cnt.idle3[0] += 64u - n_call;   // no path in this call
          cnt.idle3[1] += n_over;         // walk over, waiting for the round to end
          cnt.idle3[2] += n_parked;       // parked at a leaf
        }
      }
      if (interior) {
        if (STATS) {
          cnt.nodes++;
          if ((threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)__ballot(1)) - 1)) cnt.wave_node_iters++;
        }
        if (MODE == 1) cur = walk8_visit<STATS>(nodes4, stk, w, x8, cur, sp);
        else if (MODE == 2) cur = walk4q_visit<STATS>(nodes4, stk, w, x8.oct, cur, sp);
        else {
          long long tv0 = 0;
          if (STATS) tv0 = clock64();
          cur = walk4_visit<STATS>(nodes4, stk, w, cur, sp, &cnt);
          if (STATS) { const long long tv1 = clock64(); if (PT_WAVE_ONE()) cnt.cyc[7] += (unsigned long long)(tv1 - tv0); }   // (fetch included: [7] - [6] = arithmetic + pushes + pops)
        }
      }
      if (walkers < walk_min) break;
    }
    long long tb1 = 0;
    if (STATS) {
      tb1 = clock64();
      const bool any_leaf = __ballot(cur != PT_NONE && (cur & PT_LEAF_BIT)) != 0ull;
      if (PT_WAVE_ONE()) { cnt.cyc[1] += (unsigned long long)(tb1 - tb0); if (any_leaf) cnt.cyc[5]++; }
    }
    // leaf phase: the leaf's triangle records come from L2 — fetch them all before the first test (one round trip
    // instead of one per triangle; leaves hold at most three, longer ones finish in the plain loop)
    if (cur != PT_NONE && (cur & PT_LEAF_BIT)) {
      const uint32_t first = cur & 0xFFFFFFu, count = (cur >> 24) & 0x7Fu;
      constexpr uint32_t TS = 3u;   // float4 per triangle record
      const float4* t = tris + (size_t)first * TS;
      float4 r[9];
#pragma unroll
      for (uint32_t k = 0; k < 3u; ++k)
        if (k < count) { r[3 * k] = t[TS * k]; r[3 * k + 1] = t[TS * k + 1]; r[3 * k + 2] = t[TS * k + 2]; }
      if (!STATS && PT_ASM_LEAF_WIDE && small_det) {   // (wave-uniform)
#pragma unroll
        for (uint32_t k = 0; k < 3u; ++k)
          if (k < count) mt_test_asm(r[3 * k], r[3 * k + 1], make_float2(r[3 * k + 2].x, r[3 * k + 2].y), w.o, w.d, w.best);
      } else {
#pragma unroll
        for (uint32_t k = 0; k < 3u; ++k)
          if (k < count) {
            if (STATS) {
              const unsigned long long act = __ballot(1);
              if ((threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)act) - 1)) ++cnt.wave_tri_iters;
            }
            mt_test<false>(r[3 * k], r[3 * k + 1], r[3 * k + 2], w.o, w.d, w.best, small_det);
          }
      }
      if (STATS) cnt.tris += count < 3u ? count : 3u;
      if (count > 3u) walk_leaf<STATS>(tris, w, first + 3u, count - 3u, cnt.tris, cnt.wave_tri_iters, small_det, TS);
      cur = PT_POP_BATCH ? stack4_pop4(stk, sp, w.best.t) : stack4_pop(stk, sp, w.best.t);
    }
    if (STATS && PT_WAVE_ONE()) cnt.cyc[2] += (unsigned long long)(clock64() - tb1);
    if ((uint32_t)__popcll(__ballot(cur != PT_NONE)) < t_eff) break;
  }
  best = w.best;
}

// XCD-local regions (restart kernel, KParams::xcd_regions): workgroups are dealt round-robin to the eight XCDs and every XCD has an L2
// of its own (4 MB), so a scene walked from L2 wants each XCD to see a COMPACT part of the frame — primary rays and the first
// bounces of neighbouring pixels touch the same nodes and triangles — instead of every eighth strip of it.  Ticket T belongs to
// region T & 7 (a 4 x 2 grid of the launch's tile rectangle; heads hand out n_static + 8 t + h, so head h = XCD h serves region h
// until it runs dry and then helps the next one); entry i = T >> 3 of a region is frame i % count of its tile i / count — the
// frames of one tile run side by side on neighbouring waves and share their lines — and tiles are numbered in column strips eight
// tiles wide.  Scheduling only: a path's arithmetic does not depend on when or where it runs.
// Returns false when the region has no entry i (the ticket names no work).
PT_DEV bool region_tile(const KParams& p, uint32_t ticket, uint32_t& col, uint32_t& row, uint32_t& k)
{
  const uint32_t r = ticket & 7u, i = ticket >> 3;
  const uint32_t tiles_x = PT_KARG(p, tiles_x), tiles_y = PT_KARG(p, n_tiles) / tiles_x, count = PT_KARG(p, sample_count);
  const uint32_t rx = r & 3u, ry = r >> 2;
  const uint32_t x0 = rx * tiles_x / 4u, x1 = (rx + 1u) * tiles_x / 4u, y0 = ry * tiles_y / 2u, y1 = (ry + 1u) * tiles_y / 2u;
  const uint32_t w = x1 - x0, h = y1 - y0;
  const uint32_t j = i / count;
  if (j >= w * h) return false;
  k = i - j * count;
  const uint32_t strips = w / 8u, full = strips * 8u * h;
  if (j < full) {
    const uint32_t s = j / (8u * h), rem = j - s * 8u * h;
    row = y0 + rem / 8u;
    col = x0 + s * 8u + (rem & 7u);
  } else {
    const uint32_t rem = j - full, wl = w - strips * 8u;
    row = y0 + rem / wl;
    col = x0 + strips * 8u + rem % wl;
  }
  return true;
}

// per-wave time stamps of a launch (ptamd_set_timeline): [4 gwave + slot] = device clock at 0 kernel entry, 1 scene staged,
// 2 the wave found no ticket (tail mode from here on), 3 exit
// (an instantiation of its own, VARIANT == PT_RS_STAMPS: even switched off the four stamps cost the round loop 1.5 %)
#define PT_STAMP(slot)                                                                                                   \
  do {                                                                                                                   \
    if (VARIANT == PT_RS_STAMPS) {                                                                                       \
      unsigned long long* tl_ = PT_KARG(p, timeline);   /* (loaded by the whole wave: scalar) */                         \
      if ((threadIdx.x & 63u) == 0u)                                                                                     \
        tl_[(size_t)(blockIdx.x * (THREADS / 64u) + (threadIdx.x >> 6)) * 4u + (slot)] = wall_clock64();                 \
    }                                                                                                                    \
  } while (0)
// instantiations of the restart kernel: the shipped one, the instrumented one (counters), the one that records per-wave time
// stamps, and the one that tests every triangle instead of walking the tree (far-origin launches, ptamd_api.cpp)
#define PT_RS_PLAIN 0
#define PT_RS_STATS 1
#define PT_RS_STAMPS 2
#define PT_RS_BRUTE 3
#define PT_RS_WIDE8 4   /* scenes that do not fit in LDS walked in the eight-wide quantised form (Bvh::nodes8) instead of the four-wide one */
#define PT_RS_WIDE4Q 5  /* ... in the four-wide form with 64-byte quantised nodes (Bvh::nodes4q) */

template <bool LDS_RESIDENT, int VARIANT>
__global__ void __launch_bounds__(LDS_RESIDENT ? PT_RS_THREADS : PT_RS4_THREADS, LDS_RESIDENT ? PT_RS_WAVES_PER_EU : PT_RS4_WAVES_PER_EU)
pt_megakernel_restart(PT_KERNEL_PARAMS)
{
  constexpr bool STATS = VARIANT == PT_RS_STATS;
  constexpr uint32_t THREADS = LDS_RESIDENT ? PT_RS_THREADS : PT_RS4_THREADS;
  extern __shared__ float4 s_mem[];
  const float4* s_nodes;
  const float4* s_tris;
  PT_STAMP(0);
  stage_scene<2, LDS_RESIDENT, LDS_RESIDENT && !STATS && PT_ASM_WALK>(p, s_mem, s_nodes, s_tris);
  // scenes that do not fit in LDS are walked in the four-wide form; LDS then holds the waves' stacks
  constexpr bool WIDE = !LDS_RESIDENT;

  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t gwave = blockIdx.x * (THREADS / 64u) + (threadIdx.x >> 6);
  float4* slab = p.pool + (size_t)gwave * 192u;
  // p.pool_lds_offset != 0: the pools live in LDS behind the staged scene (576 dwords per wave) instead of the global slab
  uint32_t* lds_pool = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(s_mem) + p.pool_lds_offset) + (threadIdx.x >> 6) * 576u;
  Stack4 stk;
  stk.top = s_mem;
  stk.top_n = WIDE ? p.treelet_nodes : 0u;
  constexpr uint32_t NODE_F4 = VARIANT == PT_RS_WIDE4Q ? 4u : 8u;   // float4 per wide node
  // float4s of the treelet's LDS region: chunk-major it is 64 KB whatever the number of nodes staged (plane stride fixed at compile time)
  const uint32_t treelet_f4 = !WIDE ? 0u : (PT_TREELET_SOA ? (p.treelet_nodes ? 8u * PT_TREELET_STRIDE : 0u) : p.treelet_nodes * NODE_F4);
  if (WIDE) {
#if PT_TREELET_SOA
    constexpr uint32_t PLANE = PT_TREELET_STRIDE * (8u / NODE_F4);   // nodes per chunk plane
    for (uint32_t i = threadIdx.x; i < p.treelet_nodes * NODE_F4; i += blockDim.x) s_mem[(i % NODE_F4) * PLANE + i / NODE_F4] = p.nodes4[i];
#else
    stage_to_lds(s_mem, p.nodes4, p.treelet_nodes * NODE_F4);
#endif
    __syncthreads();
  }
  PT_STAMP(1);
  stk.lds = reinterpret_cast<uint2*>(s_mem + (size_t)treelet_f4) + (size_t)(threadIdx.x >> 6) * p.stack_lds_entries * 64u + lane;
  stk.spill = p.stack_spill + (size_t)gwave * p.stack_spill_entries * 64u + lane;
  stk.lds_entries = p.stack_lds_entries;
  uint32_t cur = PT_NONE, sp = 0;   // wide walk: reference to process next, entries on the stack
  Counters cnt = {};
  uint32_t samples = 0;

  Path st;
  st.o = st.d = st.throughput = st.acc = mk3(0.f);
  st.rng.v0 = st.rng.v1 = st.rng.v2 = st.rng.v3 = st.rng.v4 = st.rng.d = 0;
  st.specular_col = 0.f; st.xy = 0; st.bk = 0;
  bool idle = true;      // this lane has no path
  bool walking = false;  // its path is in the middle of a walk: (best, node, r1) are live
  Best best;
  best.t = PT_MAX_DIST; best.u = best.v = 0.f; best.idx = PT_END;
  uint32_t node = PT_END;
  // wave-uniform: the pool holds the paths of ONE tile, entries [pool_rd, 64) not yet handed out
  uint32_t pool_rd = 64u, tile_x0 = 0, tile_y0 = 0, tile_k = 0;
  uint32_t tile_row_delta = 0;   // (frame row) - (row of the launch's buffers + row_begin): non-zero for interleaved bands
  uint32_t tile = 0, tile_end = 0;
  // XCD-local regions (KParams::xcd_regions): the waves of XCD x take the tickets = x (mod 8), their first ones without an atomic
  uint32_t ticket = PT_KARG(p, xcd_regions) ? ((blockIdx.x >> 3) * (THREADS / 64u) + (threadIdx.x >> 6)) * 8u + (blockIdx.x & 7u) : gwave;
  ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket);   // wave-uniform: the tile arithmetic stays scalar
  uint32_t reg_col = 0, reg_row = 0, reg_k = 0;
  uint32_t head = blockIdx.x & 7u, dry = 0;
  bool have_ticket = true, exhausted = false;
  for (;;) {
    // ---- restart lanes without a path from the pool; an empty pool is refilled with a whole tile, all lanes active
    long long tr0 = 0;
    if (STATS) tr0 = clock64();
    unsigned long long need = __ballot(idle);
    while (need) {
      if (pool_rd >= 64u) {
        if (exhausted) break;
        if (tile >= tile_end) {
          const uint32_t tiles_per_ticket = PT_KARG(p, tiles_per_ticket);
          const uint32_t total = PT_KARG(p, n_tiles) * PT_KARG(p, sample_count); // (tile, frame) pairs
          // a ticket that names work: the wave's own first one (no atomic), then from this workgroup's XCD head; a head that has run
          // dry sends the wave on to the next one for good
          for (;;) {
            if (!have_ticket) {
              uint32_t* heads = PT_KARG(p, tile_heads);
              uint32_t t = 0;
              if (lane == 0) t = atomicAdd(heads + head * PT_HEAD_STRIDE, 1u);
              t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
              ticket = PT_KARG(p, n_static) + t * 8u + head;
            }
            have_ticket = false;
            if (PT_KARG(p, xcd_regions) ? region_tile(p, ticket, reg_col, reg_row, reg_k) : ((unsigned long long)ticket * tiles_per_ticket < total)) break;
            head = (head + 1u) & 7u;
            if (++dry == 8u) break;
          }
          if (dry == 8u) { exhausted = true; PT_STAMP(2); break; }
          if (PT_KARG(p, xcd_regions)) { tile = ticket; tile_end = ticket + 1u; }
          else {
            tile = ticket * tiles_per_ticket;
            tile_end = tile + tiles_per_ticket;
            if (tile_end > total) tile_end = total;
          }
        }
        {
          const uint32_t n_tiles = PT_KARG(p, n_tiles), tiles_x = PT_KARG(p, tiles_x), row_begin = PT_KARG(p, row_begin);
          uint32_t local_y0;                                      // row inside this launch's share of the frame
          if (PT_KARG(p, xcd_regions)) {
            tile_k = reg_k; tile_x0 = reg_col * PT_TILE_W; local_y0 = reg_row * PT_TILE_H;
          } else {
            tile_k = tile / n_tiles;
            const uint32_t tl = tile - tile_k * n_tiles;
            tile_x0 = (tl % tiles_x) * PT_TILE_W;
            local_y0 = (tl / tiles_x) * PT_TILE_H;
          }
          const uint32_t ilv_ranks = PT_KARG(p, ilv_ranks);
          if (ilv_ranks > 1u) {
            // interleaved bands: band j of ilv_rows rows belongs to rank j % ilv_ranks; this launch renders
            // the bands of rank ilv_rank and stores them one after the other
            const uint32_t ilv_rows = PT_KARG(p, ilv_rows);
            const uint32_t b = local_y0 / ilv_rows, within = local_y0 - b * ilv_rows;
            tile_y0 = (b * ilv_ranks + PT_KARG(p, ilv_rank)) * ilv_rows + within;
          } else {
            tile_y0 = row_begin + local_y0;
          }
          tile_row_delta = tile_y0 - (row_begin + local_y0);
        }
        ++tile;
        {
          const uint32_t x = tile_x0 + (lane & (PT_TILE_W - 1u)), y = tile_y0 + (lane >> PT_TILE_W_LOG2);
          if (x < p.width && y < p.y_limit) {
            Path fresh;
            path_begin(p, x, y, fresh, tile_k);
            if (p.pool_lds_offset) pool_store_lds(lds_pool, lane, fresh);
            else pool_store(slab, lane, fresh);
          }
          // the stores have reached L2 (LDS: are ordered before this wave's later reads) before any lane reads them back
          if (!p.pool_lds_offset) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        pool_rd = 0u;
      }
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
      const uint32_t avail = 64u - pool_rd;
      if (idle && rank < avail) {
        const uint32_t e = pool_rd + rank;
        const uint32_t x = tile_x0 + (e & (PT_TILE_W - 1u)), y = tile_y0 + (e >> PT_TILE_W_LOG2);
        if (x < p.width && y < p.y_limit) {   // entries of pixels outside the frame were never written: skip them
          if (p.pool_lds_offset) pool_load_lds(lds_pool, e, st);
          else pool_load(slab, e, st);
          st.throughput = mk3(1.0f);
          st.acc = mk3(0.0f);
          st.specular_col = 0.0f;
          // the path is done with its frame coordinates (seed and ray were made in path_begin): from here on `y` is
          // the row its sample is parked at, row_begin + the row inside the launch's buffers
          st.xy = x | ((y - tile_row_delta) << 16);
          st.bk = tile_k << 16;
          idle = false;
          walking = false;
        }
      }
      const uint32_t wanted = (uint32_t)__popcll(need);
      pool_rd += wanted < avail ? wanted : avail;
      need = __ballot(idle);
    }
    if (__ballot(!idle) == 0ull) break;
    long long tr1 = 0;
    if (STATS) { tr1 = clock64(); if (PT_WAVE_ONE()) cnt.cyc[0] += (unsigned long long)(tr1 - tr0); }

    if (!idle) {
      if (!walking) {
        best.t = PT_MAX_DIST; best.u = 0.f; best.v = 0.f; best.idx = PT_END;
        node = p.n_nodes ? 0u : PT_END;
        cur = p.n_nodes4 ? 0u : PT_NONE;
        sp = 0u;
        walking = true;
        if (STATS) cnt.rays++;
      }
      if (VARIANT == PT_RS_BRUTE) {
        // a camera too far outside the scene for the boxes' margins (ptamd_api.cpp: far_origin): every triangle record, as
        // the reference does — the (t, face index) minimum does not depend on the order they are tested in
        for (uint32_t i = 0; i < p.n_bvh_tris; ++i) mt_test<false>(s_tris[i * 3u], s_tris[i * 3u + 1u], s_tris[i * 3u + 2u], st.o, st.d, best, false);
        if (STATS) cnt.tris += p.n_bvh_tris;
        node = PT_END;
      } else if (WIDE) {
        traverse_round4<STATS, VARIANT == PT_RS_WIDE8 ? 1 : (VARIANT == PT_RS_WIDE4Q ? 2 : 0)>(p, p.nodes4, s_tris, stk, st.o, st.d, best, cur, sp, p.round_min, p.round_div, p.round_div_m16, p.walk_min4, p.small_det != 0u, cnt);
        node = cur == PT_NONE ? PT_END : 0u;
      } else {
        traverse_round<STATS, LDS_RESIDENT>(s_nodes, s_tris, p.n_nodes, st.o, st.d, best, node, p.round_min, p.round_div, p.round_div_m16, p.walk_min, p.small_det != 0u, cnt);
      }
      if (STATS) {   // fetch_events: rounds of this wave; fetch_rays: walks that completed in them
        if (lane == (uint32_t)(__ffsll((long long)__ballot(1)) - 1)) cnt.fetch_events++;
        if (node == PT_END) cnt.fetch_rays++;
      }
      if (node == PT_END) {
        // the iteration's first variate (raytrace.cu:70) is drawn here, after the search: intersect() draws nothing, so the
        // path's random stream is the same, and r1 need not live (in scratch, as it turned out) across the walk
        long long tl0 = 0;
        if (STATS) tl0 = clock64();
        const float r1 = path_pre(p, st);
        Nearest n;
        n.t = best.t; n.u = best.u; n.v = best.v; n.idx = best.idx;
        n = nearest_lights(p, st.o, st.d, n);
        walking = false;
        long long tl1 = 0;
        if (STATS) { asm volatile("" : "+v"(n.t), "+v"(n.idx)); tl1 = clock64(); if (PT_WAVE_ONE()) cnt.cyc[3] += (unsigned long long)(tl1 - tl0); }   // [3]: r1 + light loop
        if (path_post<STATS>(p, st, r1, n, cnt)) {
          path_finish_sample(p, st);
          idle = true;
          if (STATS) samples++;
        }
        if (STATS) { const long long tl2 = clock64(); if (PT_WAVE_ONE()) cnt.cyc[9] += (unsigned long long)(tl2 - tl1); }   // [9]: path_post (record fetch [8] included) + parked sample
      }
    }
    if (STATS) {   // (outside the divergent part: every lane of the wave is here)
      const long long tr3 = clock64();
      if (PT_WAVE_ONE()) cnt.cyc[4] += (unsigned long long)(tr3 - tr0);   // (light loop + shading = the rest: [4] - [0] - [1] - [2])
    }
  }
  PT_STAMP(3);
  flush_counters<STATS>(p, cnt, samples);
}

// ---------------------------------------------------------------- the megakernel, blockwise form

// Persistent WORKGROUPS with per-bounce ray compaction, octant sort and dynamic ray fetch.
//
// Measured on MI355X (DESIGN.md "Kernels"): lanes of a wave that trace incoherent bounce rays
// spend most box-test iterations masked off — with every lane live the wave runs max(length) not
// mean(length) iterations (38 % lane utilisation on bounce 1 of indoor), and paths that ended
// leave holes (19 % / 12 % on bounces 2 / 3).  Refilling holes with NEW PATHS in the middle of
// old ones (pt_megakernel_persistent) loses more to mixing coherent primary rays with incoherent
// ones than it gains.  What this kernel does instead, per super-tile of PT_BW_THREADS pixels:
//   * bounce 0 (coherent primary rays, 93 % utilisation) is traced in place;
//   * before every later bounce the live rays are written to an LDS pool in (ray octant, wave,
//     lane) order — a counting sort from 8 ballots per wave and one wave scan of the count table;
//   * the waves drain the pool cooperatively: a lane whose walk has ended stores its 16-byte
//     Nearest record and, once PT_BW_FETCH_MIN lanes of its wave are idle (ballot + popcount),
//     they take the next rays from the pool head (one LDS atomic per wave, rank = mbcnt) — the
//     wavefront-level restart keeps the box-test loop full with rays of the SAME bounce depth;
//   * owners read their record back and shade; path state never leaves its lane.
template <bool LDS_RESIDENT, bool STATS>
__global__ void __launch_bounds__(PT_BW_THREADS, PT_BW_WAVES_PER_EU) pt_megakernel_blockwise(PT_KERNEL_PARAMS)
{
  constexpr uint32_t NW = PT_BW_THREADS / 64u; // waves per workgroup
  constexpr uint32_t NE = NW * 8u;             // entries of the [octant][wave] count table
  constexpr uint32_t EPL = (NE + 63u) / 64u;   // table entries scanned per lane (1 or 2)
  static_assert(NW % 4u == 0 && NE <= 128u, "workgroup must be 4k waves, at most 16");
  extern __shared__ float4 s_mem[];
  const float4* s_nodes;
  const float4* s_tris;
  stage_scene<2, LDS_RESIDENT>(p, s_mem, s_nodes, s_tris);     // this kernel walks with the C++ loop: plain links
  // behind the staged scene: ray pool (2 float4 per slot), count table, pool head, ticket
  float4* pool = s_mem + (LDS_RESIDENT ? (p.n_nodes * 4u + p.n_bvh_tris * 3u) : 0u);
  uint32_t* s_cnt = reinterpret_cast<uint32_t*>(pool + PT_BW_THREADS * 2u);
  uint32_t* s_head = s_cnt + 128;
  uint32_t* s_ticket = s_cnt + 129;

  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  Counters cnt = {};
  uint32_t samples = 0;
  uint32_t ticket = blockIdx.x; // first super-tile is static, later ones come from the counter

  for (;;) {
    if (ticket >= p.n_tiles) break; // n_tiles counts super-tiles here (uniform across the block)
    // super-tile = 4 x NW/4 tiles of 8x8 pixels; wave w owns tile (w & 3, w >> 2)
    const uint32_t sx = (ticket % p.tiles_x) * 32u, sy = p.row_begin + (ticket / p.tiles_x) * (NW * 2u);
    const uint32_t x = sx + (wave & 3u) * 8u + (lane & 7u);
    const uint32_t y = sy + (wave >> 2) * 8u + (lane >> 3);
    const bool active = x < p.width && y < p.row_end;
    Path st;
    bool live = active;
    if (active) path_begin(p, x, y, st);
    else { st.o = st.d = st.throughput = st.acc = mk3(0.f); st.xy = 0; st.bk = 0; st.specular_col = 0.f;
           st.rng.v0 = st.rng.v1 = st.rng.v2 = st.rng.v3 = st.rng.v4 = st.rng.d = 0; }

    // bounce 0: primary rays are coherent and all lanes are live — trace in place
    if (live) live = !path_step<2, STATS>(p, s_nodes, s_tris, st, cnt);

    for (;;) {
      // ---- counting sort of the live rays by (octant, wave, lane)
      const uint32_t oct = (st.d.x < 0.f ? 1u : 0u) | (st.d.y < 0.f ? 2u : 0u) | (st.d.z < 0.f ? 4u : 0u);
      uint32_t rank = 0, my_count = 0;
#pragma unroll
      for (uint32_t o = 0; o < 8u; ++o) {
        const unsigned long long m = __ballot(live && oct == o);
        const uint32_t c = (uint32_t)__popcll(m);
        if (oct == o) rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (lane == o) my_count = c;
      }
      if (lane < 8u) s_cnt[lane * NW + wave] = my_count;
      if (tid == 0) *s_head = 0u;
      __syncthreads();
      // every wave scans the table itself (no extra barrier): lane i owns entries [i*EPL, i*EPL+EPL)
      uint32_t e0 = 0, e1 = 0;
      if (lane * EPL < NE) e0 = s_cnt[lane * EPL];
      if (EPL == 2u && lane * EPL + 1u < NE) e1 = s_cnt[lane * EPL + 1u];
      uint32_t incl = e0 + e1;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off, 64);
        if ((int)lane >= off) incl += up;
      }
      const uint32_t n_live = __shfl(incl, 63, 64);
      if (n_live == 0) break; // uniform: every wave computed the same total
      const uint32_t excl = incl - (e0 + e1);
      const uint32_t entry = oct * NW + wave;
      uint32_t base = __shfl(excl, (int)(entry / EPL), 64);
      if (EPL == 2u) { const uint32_t first = __shfl(e0, (int)(entry / EPL), 64); if (entry & 1u) base += first; }
      const uint32_t slot = base + rank;
      float r1 = 0.f;
      if (live) {
        r1 = path_pre(p, st);
        pool[slot * 2u + 0u] = make_float4(st.d.x, st.d.y, st.d.z, st.o.x);
        pool[slot * 2u + 1u] = make_float4(st.o.y, st.o.z, 0.f, 0.f);
      }
      __syncthreads();

      // ---- cooperative drain of the pool with wave-level restart
      {
        Walk w;
        w.node = PT_END; w.oct = 0; w.link_off = 8u;
        w.o = w.d = w.inv = w.noi = mk3(0.f);
        w.best.t = PT_MAX_DIST; w.best.u = w.best.v = 0.f; w.best.idx = PT_END;
        bool have = false;        // this lane is walking ray `my_ray`
        uint32_t my_ray = 0;
        bool pool_empty = false;  // wave-uniform
        for (;;) {
          const unsigned long long idle_mask = __ballot(!have);
          const uint32_t n_idle = (uint32_t)__popcll(idle_mask);
          if (!pool_empty && (n_idle >= PT_BW_FETCH_MIN || idle_mask == ~0ull)) {
            uint32_t head = 0;
            if (lane == 0) head = atomicAdd(s_head, n_idle);
            head = (uint32_t)__builtin_amdgcn_readfirstlane((int)head);
            const uint32_t r = head + __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_mask >> 32),
                                                                __builtin_amdgcn_mbcnt_lo((uint32_t)idle_mask, 0u));
            if (!have && r < n_live) {
              const float4 ra = pool[r * 2u + 0u], rb = pool[r * 2u + 1u];
              walk_init(w, mk3(ra.w, rb.x, rb.y), mk3(ra.x, ra.y, ra.z), p.n_nodes);
              have = true;
              my_ray = r;
              if (STATS) cnt.rays++;
            }
            if (head + n_idle >= n_live) pool_empty = true;
          }
          if (__ballot(have) == 0ull) break;
          if (have) {
            uint32_t leaf_first, leaf_count;
            walk_to_leaf<STATS>(s_nodes, w, leaf_first, leaf_count, cnt.nodes, cnt.wave_node_iters);
            if (leaf_count != 0u) {
              walk_leaf<STATS>(s_tris, w, leaf_first, leaf_count, cnt.tris, cnt.wave_tri_iters);
            } else if (w.node == PT_END) {
              // walk over: publish the face-search result in the ray's own slot
              pool[my_ray * 2u + 0u] = make_float4(w.best.t, w.best.u, w.best.v, u_as_f(w.best.idx));
              have = false;
            }
          }
        }
      }
      __syncthreads();
      // ---- owners pick up their record, run the light loop and shade
      if (live) {
        const float4 h = pool[slot * 2u + 0u];
        Nearest n;
        n.t = h.x; n.u = h.y; n.v = h.z; n.idx = f_as_u(h.w);
        n = nearest_lights(p, st.o, st.d, n);
        live = !path_post<STATS>(p, st, r1, n, cnt);
      }
      // the next iteration's table/head writes are ordered behind these reads by its barrier, and
      // its ray writes come after that barrier, so the pool is free again by then
    }

    if (active) { path_finish(p, st); if (STATS) samples++; }
    // next super-tile
    __syncthreads();
    if (tid == 0) *s_ticket = atomicAdd(p.tile_counter, 1u);
    __syncthreads();
    ticket = *s_ticket;
  }
  flush_counters<STATS>(p, cnt, samples);
}

// ---------------------------------------------------------------- the megakernel, split form

// Shader waves and traverser waves (producer/consumer inside one workgroup, no barriers).
//
// The box-test loop runs at 33 % lane utilisation when every lane owns one path: a wave waits for
// its longest walk, and paths that ended leave holes.  Here the walk is taken away from the path's
// lane.  A workgroup has PT_SP_SHADERS "shader" waves, which own the path state (64 paths each,
// exactly the persistent kernel's per-lane state machine) and never walk the tree, and
// PT_SP_TRAVERSERS "traverser" waves, which own no paths: they pull rays from whatever the shader
// waves have published in LDS and keep their 64 lanes busy by restarting a lane on the next ray
// as soon as PT_SP_FETCH_MIN lanes have finished (ballot + popcount, rank = mbcnt) — rays of
// different pixels, waves and bounce depths share one walk loop, which is fine because nothing
// but the (t, index) minimum of each individual ray matters.
//
// Protocol per shader wave w (all words in LDS, workgroup-scope atomics, bounded spins):
//   publish   write n rays to rays[w][0..n), done[w] = 0, then ONE store  word[w] = n << 16
//   claim     a traverser does  old = atomicAdd(word[w], k);  it owns rays
//             [old & 0xffff, min((old & 0xffff) + k, old >> 16))  — only the value the add
//             RETURNS is trusted, never an earlier read
//   result    the traverser lane overwrites its ray slot with {t, u, v, idx}, then atomicAdd(done[w], 1)
//   retire    the shader wave spins until done[w] == n, stores word[w] = 0, reads its results, shades
// LDS operations of one wave execute in order, so "write rays, then publish" and "write result,
// then count it" need no more than a compiler-level fence.
template <bool LDS_RESIDENT, bool STATS>
__global__ void __launch_bounds__(PT_SP_THREADS, PT_SP_WAVES_PER_EU) pt_megakernel_split(PT_KERNEL_PARAMS)
{
  constexpr uint32_t NW = PT_SP_THREADS / 64u, NS = PT_SP_SHADERS, NT = NW - NS;
  static_assert(NS >= 1 && NT >= 1, "need at least one shader wave and one traverser wave");
  extern __shared__ float4 s_mem[];
  const float4* s_nodes;
  const float4* s_tris;
  stage_scene<2, LDS_RESIDENT, LDS_RESIDENT && !STATS && PT_ASM_WALK>(p, s_mem, s_nodes, s_tris); // zeroes nothing: control words are set below
  float4* rays = s_mem + (LDS_RESIDENT ? (p.n_nodes * 4u + p.n_bvh_tris * 3u) : 0u); // [NS][64][2]
  uint32_t* word = reinterpret_cast<uint32_t*>(rays + NS * 64u * 2u);               // [NS] n << 16 | claimed
  uint32_t* done = word + NS;                                                          // [NS]
  uint32_t* fin = done + NS;                                                           // shader waves finished
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (tid < NS) { word[tid] = 0u; done[tid] = 0u; }
  if (tid == 0) *fin = 0u;
  __syncthreads(); // the only barrier: control words initialised (stage_scene's barrier came before the stores)

  Counters cnt = {};
  uint32_t samples = 0;
  const uint32_t spin_limit = 1u << 26; // bounded spins: a protocol bug must not hang the GPU

  if (wave < NS) {
    // ------------------------------------------------------------ shader wave
    float4* my_rays = rays + wave * 128u;
    const uint32_t total = p.n_tiles * p.sample_count;
    uint32_t ticket = blockIdx.x * NS + wave; // first tile is static, later ones come from the counter
    bool failed = false;
    while (ticket < total && !failed) {
      const uint32_t k = ticket / p.n_tiles, tl = ticket - k * p.n_tiles;
      const uint32_t x = (tl % p.tiles_x) * PT_TILE_W + (lane & (PT_TILE_W - 1u));
      const uint32_t y = p.row_begin + (tl / p.tiles_x) * PT_TILE_H + (lane >> PT_TILE_W_LOG2);
      const bool active = x < p.width && y < p.row_end;
      Path st;
      if (active) path_begin(p, x, y, st, k);
      else { st.o = st.d = st.throughput = st.acc = mk3(0.f); st.xy = 0; st.bk = 0; st.specular_col = 0.f;
             st.rng.v0 = st.rng.v1 = st.rng.v2 = st.rng.v3 = st.rng.v4 = st.rng.d = 0; }
      bool live = active;
      for (;;) {
        const unsigned long long m = __ballot(live);
        const uint32_t n = (uint32_t)__popcll(m);
        if (n == 0) break;
        const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        float r1 = 0.f;
        if (live) {
          r1 = path_pre(p, st);
          my_rays[slot * 2u + 0u] = make_float4(st.d.x, st.d.y, st.d.z, st.o.x);
          my_rays[slot * 2u + 1u] = make_float4(st.o.y, st.o.z, 0.f, 0.f);
          if (STATS) cnt.rays++;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) {
          __hip_atomic_store(&done[wave], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_store(&word[wave], n << 16, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        // wait for the traversers
        uint32_t spins = 0, d = 0;
        do {
          d = __hip_atomic_load(&done[wave], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
          d = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
          if (d != n) __builtin_amdgcn_s_sleep(2);
        } while (d != n && ++spins < spin_limit);
        if (d != n) { failed = true; break; }
        if (lane == 0) __hip_atomic_store(&word[wave], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (live) {
          const float4 h = my_rays[slot * 2u + 0u];
          Nearest nr;
          nr.t = h.x; nr.u = h.y; nr.v = h.z; nr.idx = f_as_u(h.w);
          nr = nearest_lights(p, st.o, st.d, nr);
          live = !path_post<STATS>(p, st, r1, nr, cnt);
        }
      }
      if (failed) break;
      if (active) {
        if (p.sample_count > 1u) path_finish_sample(p, st);
        else path_finish(p, st);
        if (STATS) samples++;
      }
      uint32_t t = 0;
      if (lane == 0) t = atomicAdd(p.tile_counter, 1u);
      ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    }
    if (failed && lane == 0) atomicAdd(p.error_flag, 1ull); // protocol time-out (ptamd_device_error_count)
    if (lane == 0) __hip_atomic_fetch_add(fin, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  } else {
    // ------------------------------------------------------------ traverser wave
    Walk w;
    w.node = PT_END; w.oct = 0; w.link_off = 8u;
    w.o = w.d = w.inv = w.noi = mk3(0.f);
    w.best.t = PT_MAX_DIST; w.best.u = w.best.v = 0.f; w.best.idx = PT_END;
    const uint32_t lds_nodes = (uint32_t)(uintptr_t)s_nodes;
    bool have = false;
    uint32_t my_slot = 0; // batch << 8 | ray index
    uint32_t scan = (wave - NS) % NS, idle_spins = 0;
    for (;;) {
      const unsigned long long idle_mask = __ballot(!have);
      const uint32_t n_idle = (uint32_t)__popcll(idle_mask);
      if (n_idle >= PT_SP_FETCH_MIN || idle_mask == ~0ull) {
        // look for a batch with unclaimed rays, starting where the last one was found
        uint32_t got = 0, base = 0, b = scan;
        for (uint32_t tries = 0; tries < NS && got == 0; ++tries) {
          uint32_t wd = __hip_atomic_load(&word[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          wd = (uint32_t)__builtin_amdgcn_readfirstlane((int)wd);
          if ((wd & 0xffffu) < (wd >> 16)) {
            uint32_t old = 0;
            if (lane == 0) old = __hip_atomic_fetch_add(&word[b], n_idle, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            old = (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
            const uint32_t on = old >> 16, oc = old & 0xffffu;
            if (oc < on) { base = oc; got = on - oc < n_idle ? on - oc : n_idle; }
          }
          if (got == 0) b = b + 1u == NS ? 0u : b + 1u;
        }
        scan = b;
        if (got) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_mask >> 32),
                                                          __builtin_amdgcn_mbcnt_lo((uint32_t)idle_mask, 0u));
          if (!have && rank < got) {
            const uint32_t r = base + rank;
            const float4 ra = rays[(b * 64u + r) * 2u + 0u], rb = rays[(b * 64u + r) * 2u + 1u];
            walk_init(w, mk3(ra.w, rb.x, rb.y), mk3(ra.x, ra.y, ra.z), p.n_nodes, LDS_RESIDENT && !STATS && PT_ASM_WALK);
            have = true;
            my_slot = (b << 8) | r;
          }
          idle_spins = 0;
          if (STATS && lane == 0) { cnt.fetch_events++; cnt.fetch_rays += got; }
        } else if (idle_mask == ~0ull) {
          // nothing to do: leave when every shader wave is finished, else nap and look again
          uint32_t f = __hip_atomic_load(fin, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
          f = (uint32_t)__builtin_amdgcn_readfirstlane((int)f);
          if (f == NS || ++idle_spins >= spin_limit) break;
          __builtin_amdgcn_s_sleep(2);
          continue;
        }
      }
      if (have) {
        uint32_t leaf_first, leaf_count;
        if (LDS_RESIDENT && !STATS && PT_ASM_WALK) walk_to_leaf_lds(lds_nodes, w, leaf_first, leaf_count);
        else walk_to_leaf<STATS>(s_nodes, w, leaf_first, leaf_count, cnt.nodes, cnt.wave_node_iters);
        if (leaf_count != 0u) {
          walk_leaf<STATS>(s_tris, w, leaf_first, leaf_count, cnt.tris, cnt.wave_tri_iters);
        } else if (w.node == PT_END) {
          const uint32_t b = my_slot >> 8, r = my_slot & 0xffu;
          rays[(b * 64u + r) * 2u + 0u] = make_float4(w.best.t, w.best.u, w.best.v, u_as_f(w.best.idx));
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __hip_atomic_fetch_add(&done[b], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
          have = false;
        }
      }
    }
  }
  flush_counters<STATS>(p, cnt, samples);
}

// ---------------------------------------------------------------- gamma table: construction and exhaustive check

// thread j (1..256) finds T[j]: the smallest non-negative binary32 x (by bit pattern) with gamma_value_exact(x) >= j
__global__ void __launch_bounds__(320) pt_build_gamma_table(float* T)
{
  const uint32_t j = threadIdx.x;
  if (j > 256u) return;
  if (j == 0u) { T[0] = 0.0f; T[257] = 0.0f; return; }
  uint32_t lo = 0u, hi = 0x7F800000u;          // value(lo) < j <= value(hi): +0 gives 0, +inf saturates
  while (hi - lo > 1u) {
    const uint32_t mid = lo + (hi - lo) / 2u;
    if (gamma_value_exact(u_as_f(mid)) >= j) hi = mid; else lo = mid;
  }
  T[j] = u_as_f(hi);
}

// every positive binary32 below T[256] (bit patterns first .. first + count - 1): table form against the pt_powf form
__global__ void __launch_bounds__(256) pt_gamma_selftest(const float* T, uint32_t first, uint32_t count, unsigned long long* out)
{
  __shared__ float s_T[258];
  for (uint32_t i = threadIdx.x; i < 258u; i += 256u) s_T[i] = T[i];
  __syncthreads();
  unsigned long long bad = 0;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < count; i += gridDim.x * 256u) {
    const float x = u_as_f(first + i);
    if (gamma_byte(x, s_T) != gamma_byte_exact(x)) ++bad;
  }
  if (bad) atomicAdd(out, bad);
}

// Second kernel of a batched launch: per pixel, apply the parked samples to the temporal
// framebuffer in frame order — t = t * is_static + s_k for k = 0..count-1, exactly what `count`
// consecutive launches do (raytrace.cu:255-256, is_static == 1) — then tonemap once with the last
// frame number (the intermediate surfaces of frames 0..count-2 would be overwritten anyway).
__global__ void __launch_bounds__(256) pt_resolve_kernel(PT_KERNEL_PARAMS)
{
  // the ticket heads of this launch are spent (the megakernel has finished: stream order): leave them zeroed for the
  // launch that gets this slot of the ring next, instead of a memset in front of every launch
  if (blockIdx.x == 0 && threadIdx.x < 8u && p.tile_heads) p.tile_heads[threadIdx.x * PT_HEAD_STRIDE] = 0u;
  const uint32_t rows = p.row_end - p.row_begin;
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= rows * p.width) return;
  const uint32_t x = i % p.width, y = p.row_begin + i / p.width;
  const size_t ti = (size_t)(p.height - y - 1u - p.tfb_row0) * p.width + x;
  float* tp = p.tfb + ti * 3;
  f3 t = p.tfb_reset ? mk3(0.0f) : mk3(tp[0], tp[1], tp[2]);   // tfb_reset: the accumulator counts as freshly zeroed
  for (uint32_t k = 0; k < p.sample_count; ++k) {
    const float* sp = p.samples_out + (((size_t)k * rows + (y - p.row_begin)) * p.width + x) * 3;
    t = t * (float)p.is_static;
    t = t + mk3(sp[0], sp[1], sp[2]);
  }
  tp[0] = t.x; tp[1] = t.y; tp[2] = t.z;
  f3 rad = p.frame_nb_inv != 0.0f ? t * p.frame_nb_inv : t / p.frame_nb_f; // (float)(frame_nb0 + count - 1); KParams::frame_nb_inv
  rad = exposure(rad);
  uint32_t px;
  if (p.gamma_table && p.post_id == 0u) {
    px = gamma_byte(rad.x, p.gamma_table) | (gamma_byte(rad.y, p.gamma_table) << 8) | (gamma_byte(rad.z, p.gamma_table) << 16);
  } else {
    const float g = 1.0f / 2.2f;
    rad = mk3(pt_powf(rad.x, g), pt_powf(rad.y, g), pt_powf(rad.z, g));
    rad = post_process(p.post_id, rad);
    px = (pt_f2u(rad.x * 255.0f) & 0xffu) | ((pt_f2u(rad.y * 255.0f) & 0xffu) << 8) | ((pt_f2u(rad.z * 255.0f) & 0xffu) << 16);
  }
  p.surface[(size_t)(y - p.surf_row0) * p.width + x] = px;
}

// The same pass, four horizontally adjacent pixels per thread: the 48 bytes of a pixel group are three float4s in the
// accumulator and in every sample plane, the four RGBA8 results one uint4.  Per-pixel arithmetic is unchanged.
__global__ void __launch_bounds__(256) pt_resolve_kernel4(PT_KERNEL_PARAMS)
{
  // the ticket heads of this launch are spent (the megakernel has finished: stream order): leave them zeroed for the
  // launch that gets this slot of the ring next, instead of a memset in front of every launch
  if (blockIdx.x == 0 && threadIdx.x < 8u && p.tile_heads) p.tile_heads[threadIdx.x * PT_HEAD_STRIDE] = 0u;
  __shared__ float s_gamma[258];
  const bool use_table = p.gamma_table != nullptr && p.post_id == 0u;
  if (use_table) {
    for (uint32_t i = threadIdx.x; i < 258u; i += 256u) s_gamma[i] = p.gamma_table[i];
    __syncthreads();
  }
  const uint32_t rows = p.row_end - p.row_begin;
  const uint32_t groups_per_row = p.width / 4u;
  const uint32_t g = blockIdx.x * 256u + threadIdx.x;
  if (g >= rows * groups_per_row) return;
  const uint32_t x0 = (g % groups_per_row) * 4u, y = p.row_begin + g / groups_per_row;
  const size_t ti = (size_t)(p.height - y - 1u - p.tfb_row0) * p.width + x0;
  float4* tp = reinterpret_cast<float4*>(p.tfb + ti * 3);
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a, c = a;
  if (!p.tfb_reset) { a = tp[0]; b = tp[1]; c = tp[2]; }   // tfb_reset: the accumulator counts as freshly zeroed
  float t[12] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w };
  const float is_static = (float)p.is_static;
  for (uint32_t k = 0; k < p.sample_count; ++k) {
    const float4* sp = reinterpret_cast<const float4*>(p.samples_out + (((size_t)k * rows + (y - p.row_begin)) * p.width + x0) * 3);
    const float4 s0 = sp[0], s1 = sp[1], s2 = sp[2];
    const float sv[12] = { s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w, s2.x, s2.y, s2.z, s2.w };
    for (int i = 0; i < 12; ++i) { t[i] = t[i] * is_static; t[i] = t[i] + sv[i]; }
  }
  tp[0] = make_float4(t[0], t[1], t[2], t[3]); tp[1] = make_float4(t[4], t[5], t[6], t[7]); tp[2] = make_float4(t[8], t[9], t[10], t[11]);
  uint32_t px[4];
  const float g22 = 1.0f / 2.2f;
  for (int q = 0; q < 4; ++q) {
    const f3 tq = mk3(t[q * 3 + 0], t[q * 3 + 1], t[q * 3 + 2]);
    f3 rad = p.frame_nb_inv != 0.0f ? tq * p.frame_nb_inv : tq / p.frame_nb_f;   // KParams::frame_nb_inv
    rad = exposure(rad);
    if (use_table) {
      px[q] = gamma_byte(rad.x, s_gamma) | (gamma_byte(rad.y, s_gamma) << 8) | (gamma_byte(rad.z, s_gamma) << 16);
      continue;
    }
    rad = mk3(pt_powf(rad.x, g22), pt_powf(rad.y, g22), pt_powf(rad.z, g22));
    rad = post_process(p.post_id, rad);
    px[q] = (pt_f2u(rad.x * 255.0f) & 0xffu) | ((pt_f2u(rad.y * 255.0f) & 0xffu) << 8) | ((pt_f2u(rad.z * 255.0f) & 0xffu) << 16);
  }
  *reinterpret_cast<uint4*>(p.surface + (size_t)(y - p.surf_row0) * p.width + x0) = make_uint4(px[0], px[1], px[2], px[3]);
}

// The same query through the four-wide walk: one wave per block, the stack entirely in (dynamic) LDS.
template <int MODE>
__global__ void __launch_bounds__(64) pt_trace_rays_wide_kernel(PT_KERNEL_PARAMS, const float* rays, uint32_t n, int4* out)
{
  extern __shared__ float4 s_mem[];
  const uint32_t i = blockIdx.x * 64u + threadIdx.x;
  const bool live = i < n;
  Stack4 stk;
  stk.lds = reinterpret_cast<uint2*>(s_mem) + threadIdx.x;
  stk.spill = p.stack_spill;   // never reached: the host sizes the LDS part for the whole stack
  stk.lds_entries = p.stack_lds_entries;
  stk.top = s_mem;
  stk.top_n = 0u;
  f3 d = mk3(0.f, 0.f, 1.f), o = mk3(0.f);
  if (live) { d = mk3(rays[i * 6 + 0], rays[i * 6 + 1], rays[i * 6 + 2]); o = mk3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]); }
  Counters cnt = {};
  Best best;
  best.t = PT_MAX_DIST; best.u = best.v = 0.f; best.idx = PT_END;
  uint32_t cur = (live && p.n_nodes4) ? 0u : PT_NONE, sp = 0u;
  if (live) traverse_round4<false, MODE>(p, p.nodes4, p.tris_bvh, stk, o, d, best, cur, sp, 1u, 64u, 1024u, 1u, p.small_det != 0u, cnt);
  if (!live) return;
  Nearest nr;
  nr.t = best.t; nr.u = best.u; nr.v = best.v; nr.idx = best.idx;
  nr = nearest_lights(p, o, d, nr);
  const int kind = nr.idx == PT_END ? 0 : ((nr.idx & PT_LIGHT) ? 2 : 1);
  const int index = kind == 0 ? -1 : (int)(nr.idx & ~PT_LIGHT);
  out[i] = make_int4(kind, index, (int)f_as_u(nr.t), 0);
}

// ---------------------------------------------------------------- walk-only kernel fed from a ray queue (round 4)
//
// The four-wide walk WITHOUT a path around it: persistent waves pull rays {dir, origin} from a global queue, walk them with the
// same visit code as the restart kernel (fetch_node4 / walk4_select_asm, chunk-major LDS treelet, per-lane LDS stack with a global
// continuation) and write {kind, index, t bits} records.  A lane whose walk ends takes the next ray as soon as `refill_min` lanes
// of its wave are idle: no rounds, no shading, no path state — ~20 VGPRs less than the restart kernel, so the SAME walk can be
// measured at 4, 5 and 6 waves per SIMD (template parameters: workgroup size, waves per SIMD the register budget aims for, nodes
// per treelet plane).  A measurement hook (ptamd_trace_rays_queue; scripts/gpu_trace_queue.py; profiles/r04_notes.md): it is what
// showed that the LDS part of the stack, not the number of waves, was what the wide walk was short of.
template <int THREADS, int WPE, uint32_t PLANE_NODES>
__global__ void __launch_bounds__(THREADS, WPE) pt_trace_queue_kernel(PT_KERNEL_PARAMS, const float* rays, uint32_t n, int4* out, uint32_t* head)
{
  extern __shared__ float4 s_mem[];
  constexpr uint32_t PLANE_BYTES = PLANE_NODES * 16u;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t gwave = blockIdx.x * (THREADS / 64u) + (threadIdx.x >> 6);
  for (uint32_t i = threadIdx.x; i < p.treelet_nodes * 8u; i += THREADS) s_mem[(i & 7u) * PLANE_NODES + (i >> 3)] = p.nodes4[i];
  __syncthreads();
  Stack4 stk;
  stk.top = s_mem;
  stk.top_n = p.treelet_nodes;
  stk.lds = reinterpret_cast<uint2*>(s_mem + (p.treelet_nodes ? 8u * PLANE_NODES : 0u)) + (size_t)(threadIdx.x >> 6) * p.stack_lds_entries * 64u + lane;
  stk.spill = p.stack_spill + (size_t)gwave * p.stack_spill_entries * 64u + lane;
  stk.lds_entries = p.stack_lds_entries;
  Counters cnt = {};
  Walk w;
  walk_init(w, mk3(0.f), mk3(0.f, 0.f, 1.f), 1u);
  uint32_t cur = PT_NONE, sp = 0u, ray = PT_NONE;
  uint32_t chunk_next = 0u, chunk_end = 0u;   // wave-uniform: the rays [chunk_next, chunk_end) of the queue are this wave's to hand out
  bool exhausted = false;
  const bool small_det = false;   // caller-supplied directions need not be unit vectors
  for (;;) {
    // ---- lanes without a ray take the next ones of the wave's chunk of the queue (one atomic per 512 rays: a single counter
    // hands out ~88 tickets per microsecond chip-wide, MI355X_MICROARCH.md — per-refill atomics capped the kernel at 2.8 Grays/s)
    const unsigned long long idle = __ballot(ray == PT_NONE);
    const uint32_t n_idle = (uint32_t)__popcll(idle);
    if (!exhausted && n_idle != 0u && (n_idle >= p.refill_min || n_idle == 64u)) {
      if (chunk_next >= chunk_end) {
        uint32_t base = 0u;
        if (lane == 0u) base = atomicAdd(head, 512u);
        chunk_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        chunk_end = chunk_next + 512u < n ? chunk_next + 512u : n;
        if (chunk_next >= n) { exhausted = true; chunk_next = chunk_end = n; }
      }
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
      if (ray == PT_NONE && chunk_next + rank < chunk_end) {
        ray = chunk_next + rank;
        const float* r = rays + (size_t)ray * 6u;
        walk_init(w, mk3(r[3], r[4], r[5]), mk3(r[0], r[1], r[2]), 1u);
        cur = p.n_nodes4 ? 0u : PT_NONE;
        sp = 0u;
      }
      chunk_next = chunk_next + n_idle < chunk_end ? chunk_next + n_idle : chunk_end;
    }
    if (__ballot(ray != PT_NONE) == 0ull) break;
    if (ray != PT_NONE) {
      w.oct_x = __ballot((w.oct & 1u) != 0u); w.oct_y = __ballot((w.oct & 2u) != 0u); w.oct_z = __ballot((w.oct & 4u) != 0u);
      // box phase
      for (;;) {
        const bool interior = cur < PT_LEAF_BIT;
        const uint32_t walkers = (uint32_t)__popcll(__ballot(interior));
        if (walkers == 0u) break;
        if (interior) cur = walk4_visit<false, PLANE_BYTES>(p.nodes4, stk, w, cur, sp);
        if (walkers < p.walk_min4) break;
      }
      // leaf phase
      if (cur != PT_NONE && (cur & PT_LEAF_BIT)) {
        const uint32_t first = cur & 0xFFFFFFu, count = (cur >> 24) & 0x7Fu;
        const float4* t = p.tris_bvh + (size_t)first * 3u;
        float4 r[9];
#pragma unroll
        for (uint32_t k = 0; k < 3u; ++k)
          if (k < count) { r[3 * k] = t[3 * k]; r[3 * k + 1] = t[3 * k + 1]; r[3 * k + 2] = t[3 * k + 2]; }
#pragma unroll
        for (uint32_t k = 0; k < 3u; ++k)
          if (k < count) mt_test<false>(r[3 * k], r[3 * k + 1], r[3 * k + 2], w.o, w.d, w.best, small_det);
        if (count > 3u) walk_leaf<false>(p.tris_bvh, w, first + 3u, count - 3u, cnt.tris, cnt.wave_tri_iters, small_det);
        cur = PT_POP_BATCH ? stack4_pop4(stk, sp, w.best.t) : stack4_pop(stk, sp, w.best.t);
      }
      if (cur == PT_NONE) {   // the walk is over: light spheres, the record, and the lane is free again
        Nearest nr;
        nr.t = w.best.t; nr.u = w.best.u; nr.v = w.best.v; nr.idx = w.best.idx;
        nr = nearest_lights(p, w.o, w.d, nr);
        const int kind = nr.idx == PT_END ? 0 : ((nr.idx & PT_LIGHT) ? 2 : 1);
        const int index = kind == 0 ? -1 : (int)(nr.idx & ~PT_LIGHT);
        out[ray] = make_int4(kind, index, (int)f_as_u(nr.t), 0);
        ray = PT_NONE;
      }
    }
  }
}

// Nearest-hit query on explicit rays (tests: BVH vs brute force on the device).
template <int KIND>
__global__ void __launch_bounds__(256) pt_trace_rays_kernel(PT_KERNEL_PARAMS, const float* rays, uint32_t n, int4* out)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const f3 d = mk3(rays[i * 6 + 0], rays[i * 6 + 1], rays[i * 6 + 2]);
  const f3 o = mk3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]);
  Counters cnt = {};
  const Nearest nr = trace_nearest<KIND, false>(p, p.nodes, KIND == 1 ? p.tris_brute : p.tris_bvh, o, d, cnt);
  const int kind = nr.idx == PT_END ? 0 : ((nr.idx & PT_LIGHT) ? 2 : 1);
  const int index = kind == 0 ? -1 : (int)(nr.idx & ~PT_LIGHT);
  out[i] = make_int4(kind, index, (int)f_as_u(nr.t), 0);
}

// ---------------------------------------------------------------- launchers

#ifndef PT_FMA_BUILD   /* (the contracted instantiation, pt_kernels_fma.hip, only carries the restart kernel) */
template <int KIND, bool LDS_RES, bool STATS>
static hipError_t launch_variant(const KParams& p, dim3 grid, size_t lds_bytes, hipStream_t stream)
{
  auto kern = pt_megakernel<KIND, LDS_RES, STATS, PT_TILE_THREADS>;
  if (lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, grid, dim3(PT_TILE_THREADS), lds_bytes, stream, p);
  return hipGetLastError();
}

hipError_t launch_megakernel(const KParams& p, int kind, bool lds_resident, size_t lds_bytes, bool stats,
                             hipStream_t stream)
{
  const uint32_t rows = p.row_end - p.row_begin;
  if (rows == 0 || p.width == 0) return hipSuccess;
  const uint32_t block_rows = PT_TILE_THREADS / 16u;
  dim3 grid((p.width + 15u) / 16u, (rows + block_rows - 1u) / block_rows);
  if (!lds_resident) lds_bytes = 0;
#define PT_DISPATCH(K, L, S) return launch_variant<K, L, S>(p, grid, lds_bytes, stream)
  if (kind == 1) {
    if (lds_resident) { if (stats) PT_DISPATCH(1, true, true); else PT_DISPATCH(1, true, false); }
    else { if (stats) PT_DISPATCH(1, false, true); else PT_DISPATCH(1, false, false); }
  } else {
    if (lds_resident) { if (stats) PT_DISPATCH(2, true, true); else PT_DISPATCH(2, true, false); }
    else { if (stats) PT_DISPATCH(2, false, true); else PT_DISPATCH(2, false, false); }
  }
#undef PT_DISPATCH
}

template <bool LDS_RES, bool STATS>
static const void* persistent_entry()
{
  return reinterpret_cast<const void*>(pt_megakernel_persistent<2, LDS_RES, STATS>);
}

static const void* persistent_select(bool lds_resident, bool stats)
{
  if (lds_resident) return stats ? persistent_entry<true, true>() : persistent_entry<true, false>();
  return stats ? persistent_entry<false, true>() : persistent_entry<false, false>();
}

// Resident workgroups per CU of the persistent variant (sizes its grid).
hipError_t persistent_blocks_per_cu(bool lds_resident, size_t lds_bytes, int* out)
{
  if (!lds_resident) lds_bytes = 0;
  const void* fn = persistent_select(lds_resident, false);
  if (lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
  }
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(out, fn, PT_PERSISTENT_THREADS, lds_bytes);
}

hipError_t launch_megakernel_persistent(const KParams& p, bool lds_resident, size_t lds_bytes, bool stats,
                                        uint32_t n_blocks, hipStream_t stream)
{
  if (p.n_tiles == 0 || n_blocks == 0) return hipSuccess;
  if (!lds_resident) lds_bytes = 0;
  const void* fn = persistent_select(lds_resident, stats);
  if (lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
  }
  KParams pc = p;
  void* args[] = { &pc };
  return hipLaunchKernel(fn, dim3(n_blocks), dim3(PT_PERSISTENT_THREADS), args, lds_bytes, stream);
}

#endif

template <bool LDS_RES>
static const void* restart_entry(int variant)
{
  switch (variant) {
#ifdef PT_FMA_BUILD
    case PT_RS_BRUTE: return reinterpret_cast<const void*>(pt_megakernel_restart<LDS_RES, PT_RS_BRUTE>);
    default: return reinterpret_cast<const void*>(pt_megakernel_restart<LDS_RES, PT_RS_PLAIN>);
  }
}
template <bool LDS_RES>
static const void* restart_entry_unused(int variant)
{
  switch (variant) {
#endif
    case PT_RS_STATS: return reinterpret_cast<const void*>(pt_megakernel_restart<LDS_RES, PT_RS_STATS>);
    case PT_RS_STAMPS: return reinterpret_cast<const void*>(pt_megakernel_restart<LDS_RES, PT_RS_STAMPS>);
    case PT_RS_BRUTE: return reinterpret_cast<const void*>(pt_megakernel_restart<LDS_RES, PT_RS_BRUTE>);
    case PT_RS_WIDE8: return reinterpret_cast<const void*>(pt_megakernel_restart<false, PT_RS_WIDE8>);   // (only ever a non-resident scene)
    case PT_RS_WIDE4Q: return reinterpret_cast<const void*>(pt_megakernel_restart<false, PT_RS_WIDE4Q>);
    default: return reinterpret_cast<const void*>(pt_megakernel_restart<LDS_RES, PT_RS_PLAIN>);
  }
}

// variant: instrumented build when counters are wanted, else the far-origin form, else the time-stamp form, else the shipped kernel
static const void* restart_select(bool lds_resident, bool stats, const KParams* p = nullptr)
{
  const int variant = stats ? PT_RS_STATS : (p && p->brute_walk ? PT_RS_BRUTE : (p && p->timeline ? PT_RS_STAMPS : (p && p->wide8 && !lds_resident ? (p->wide8 == 2u ? PT_RS_WIDE4Q : PT_RS_WIDE8) : PT_RS_PLAIN)));
  return lds_resident ? restart_entry<true>(variant) : restart_entry<false>(variant);
}

// chunk-major treelet (PT_TREELET_SOA): bytes of its LDS region and the most 128-byte nodes it can hold (0: node-major, sized by the node count)
uint32_t restart_treelet_region_bytes() { return PT_TREELET_SOA ? 8u * PT_TREELET_STRIDE * 16u : 0u; }
uint32_t restart_threads(bool lds_resident) { return lds_resident ? PT_RS_THREADS : PT_RS4_THREADS; }
// wide walk: resident workgroups per CU the launch bounds aim for (their LDS share holds the treelet and the waves' stacks)
uint32_t restart_wide_blocks_per_cu() { return (PT_RS4_WAVES_PER_EU * 256u) / PT_RS4_THREADS; }

// lds_bytes: the staged scene when lds_resident, else the stacks of the wide walk
hipError_t restart_blocks_per_cu(bool lds_resident, size_t lds_bytes, int* out)
{
  const void* fn = restart_select(lds_resident, false);
  if (lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
  }
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(out, fn, (int)restart_threads(lds_resident), lds_bytes);
}

hipError_t launch_megakernel_restart(const KParams& p, bool lds_resident, size_t lds_bytes, bool stats,
                                     uint32_t n_blocks, hipStream_t stream)
{
  if (p.n_tiles == 0 || n_blocks == 0) return hipSuccess;
  const void* fn = restart_select(lds_resident, stats, &p);
  if (lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
  }
  KParams pc = p;
  void* args[] = { &pc };
  return hipLaunchKernel(fn, dim3(n_blocks), dim3(restart_threads(lds_resident)), args, lds_bytes, stream);
}

#ifndef PT_FMA_BUILD
static const void* split_select(bool lds_resident, bool stats)
{
  if (lds_resident)
    return stats ? reinterpret_cast<const void*>(pt_megakernel_split<true, true>)
                 : reinterpret_cast<const void*>(pt_megakernel_split<true, false>);
  return stats ? reinterpret_cast<const void*>(pt_megakernel_split<false, true>)
               : reinterpret_cast<const void*>(pt_megakernel_split<false, false>);
}

// LDS of the split variant = staged scene (when resident) + ray batches of the shader waves + control words
size_t split_lds_bytes(bool lds_resident, size_t scene_lds_bytes)
{
  return (lds_resident ? scene_lds_bytes : 0) + (size_t)PT_SP_SHADERS * 64u * 32u + (2u * PT_SP_SHADERS + 4u) * 4u;
}

uint32_t split_shader_waves() { return PT_SP_SHADERS; }

hipError_t split_blocks_per_cu(bool lds_resident, size_t scene_lds_bytes, int* out)
{
  const size_t lds = split_lds_bytes(lds_resident, scene_lds_bytes);
  const void* fn = split_select(lds_resident, false);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(out, fn, PT_SP_THREADS, lds);
}

hipError_t launch_megakernel_split(const KParams& p, bool lds_resident, size_t scene_lds_bytes, bool stats,
                                   uint32_t n_blocks, hipStream_t stream)
{
  if (p.n_tiles == 0 || n_blocks == 0) return hipSuccess;
  const size_t lds = split_lds_bytes(lds_resident, scene_lds_bytes);
  const void* fn = split_select(lds_resident, stats);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  KParams pc = p;
  void* args[] = { &pc };
  return hipLaunchKernel(fn, dim3(n_blocks), dim3(PT_SP_THREADS), args, lds, stream);
}

static const void* blockwise_select(bool lds_resident, bool stats)
{
  if (lds_resident)
    return stats ? reinterpret_cast<const void*>(pt_megakernel_blockwise<true, true>)
                 : reinterpret_cast<const void*>(pt_megakernel_blockwise<true, false>);
  return stats ? reinterpret_cast<const void*>(pt_megakernel_blockwise<false, true>)
               : reinterpret_cast<const void*>(pt_megakernel_blockwise<false, false>);
}

// LDS of the blockwise variant = staged scene (when resident) + exchange area + count table + ticket
size_t blockwise_lds_bytes(bool lds_resident, size_t scene_lds_bytes)
{
  return (lds_resident ? scene_lds_bytes : 0) + (size_t)PT_BW_THREADS * 32u + 128u * 4u + 16u;
}

hipError_t blockwise_blocks_per_cu(bool lds_resident, size_t scene_lds_bytes, int* out)
{
  const size_t lds = blockwise_lds_bytes(lds_resident, scene_lds_bytes);
  const void* fn = blockwise_select(lds_resident, false);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(out, fn, PT_BW_THREADS, lds);
}

hipError_t launch_megakernel_blockwise(const KParams& p, bool lds_resident, size_t scene_lds_bytes, bool stats,
                                       uint32_t n_blocks, hipStream_t stream)
{
  if (p.n_tiles == 0 || n_blocks == 0) return hipSuccess;
  const size_t lds = blockwise_lds_bytes(lds_resident, scene_lds_bytes);
  const void* fn = blockwise_select(lds_resident, stats);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  KParams pc = p;
  void* args[] = { &pc };
  return hipLaunchKernel(fn, dim3(n_blocks), dim3(PT_BW_THREADS), args, lds, stream);
}

hipError_t launch_resolve(const KParams& p, hipStream_t stream)
{
  const uint32_t n = (p.row_end - p.row_begin) * p.width;
  if (n == 0) return hipSuccess;
  // four pixels per thread with 16-byte accesses when rows are whole groups of four pixels and every row starts
  // 16-byte aligned (48 bytes per group; hipMalloc'd buffers and band offsets in whole rows guarantee the bases)
  const bool wide = (p.width % 4u) == 0u && ((uintptr_t)p.tfb % 16u) == 0u && ((uintptr_t)p.samples_out % 16u) == 0u &&
                    ((uintptr_t)p.surface % 16u) == 0u;
  if (wide) hipLaunchKernelGGL(pt_resolve_kernel4, dim3((n / 4u + 255u) / 256u), dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(pt_resolve_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, p);
  return hipGetLastError();
}

hipError_t build_gamma_table(float* table_dev, hipStream_t stream)
{
  hipLaunchKernelGGL(pt_build_gamma_table, dim3(1), dim3(320), 0, stream, table_dev);
  return hipGetLastError();
}

hipError_t launch_gamma_selftest(const float* table_dev, uint32_t first, uint32_t count, unsigned long long* out_dev, hipStream_t stream)
{
  if (count == 0) return hipSuccess;
  hipLaunchKernelGGL(pt_gamma_selftest, dim3(4096), dim3(256), 0, stream, table_dev, first, count, out_dev);
  return hipGetLastError();
}

hipError_t resolve_kernels()
{
  const void* fns[] = {
    persistent_select(true, false), persistent_select(false, false), persistent_select(true, true), persistent_select(false, true),
    split_select(true, false), split_select(false, false), blockwise_select(true, false), blockwise_select(false, false),
    restart_entry<true>(PT_RS_PLAIN), restart_entry<false>(PT_RS_PLAIN), restart_entry<true>(PT_RS_STATS), restart_entry<false>(PT_RS_STATS),
    restart_entry<true>(PT_RS_STAMPS), restart_entry<false>(PT_RS_STAMPS), restart_entry<true>(PT_RS_BRUTE), restart_entry<false>(PT_RS_BRUTE), restart_entry<false>(PT_RS_WIDE8), restart_entry<false>(PT_RS_WIDE4Q),
    reinterpret_cast<const void*>(pt_megakernel<1, true, false, PT_TILE_THREADS>),
    reinterpret_cast<const void*>(pt_megakernel<1, false, false, PT_TILE_THREADS>),
    reinterpret_cast<const void*>(pt_megakernel<2, true, false, PT_TILE_THREADS>),
    reinterpret_cast<const void*>(pt_megakernel<2, false, false, PT_TILE_THREADS>),
    reinterpret_cast<const void*>(pt_resolve_kernel), reinterpret_cast<const void*>(pt_resolve_kernel4),
    reinterpret_cast<const void*>(pt_trace_rays_kernel<1>), reinterpret_cast<const void*>(pt_trace_rays_kernel<2>),
    reinterpret_cast<const void*>(pt_trace_rays_wide_kernel<0>), reinterpret_cast<const void*>(pt_trace_rays_wide_kernel<1>),
    reinterpret_cast<const void*>(pt_trace_rays_wide_kernel<2>),
  };
  for (const void* fn : fns) {
    hipFuncAttributes attr;
    hipError_t e = hipFuncGetAttributes(&attr, fn);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// config: 0 = 16 waves x 1 workgroup per CU (4 waves per SIMD) with a 512-node treelet, 1 = 10 x 2 (5) with 256 nodes each,
// 2 = 12 x 2 (6) with 256 nodes each, 3 = 16 x 1 (4) with a 256-node treelet (the treelet's share of the difference)
static const void* trace_queue_entry(uint32_t config, uint32_t* threads, uint32_t* plane_nodes, uint32_t* blocks_per_cu)
{
  switch (config) {
    case 1: *threads = 640; *plane_nodes = 256; *blocks_per_cu = 2; return reinterpret_cast<const void*>(pt_trace_queue_kernel<640, 5, 256>);
    case 2: *threads = 768; *plane_nodes = 256; *blocks_per_cu = 2; return reinterpret_cast<const void*>(pt_trace_queue_kernel<768, 6, 256>);
    case 3: *threads = 1024; *plane_nodes = 256; *blocks_per_cu = 1; return reinterpret_cast<const void*>(pt_trace_queue_kernel<1024, 4, 256>);
    default: *threads = 1024; *plane_nodes = 512; *blocks_per_cu = 1; return reinterpret_cast<const void*>(pt_trace_queue_kernel<1024, 4, 512>);
  }
}

void trace_queue_shape(uint32_t config, uint32_t* threads, uint32_t* plane_nodes, uint32_t* blocks_per_cu)
{
  (void)trace_queue_entry(config, threads, plane_nodes, blocks_per_cu);
}

hipError_t launch_trace_queue(const KParams& p, uint32_t config, size_t lds_bytes, uint32_t n_blocks, const float* rays_dev, uint32_t n, int4* out_dev,
                              uint32_t* head_dev, int* resident_blocks_per_cu, hipStream_t stream)
{
  uint32_t threads, plane, bpc;
  const void* fn = trace_queue_entry(config, &threads, &plane, &bpc);
  if (resident_blocks_per_cu && lds_bytes > 64 * 1024) {   // (first launch of a configuration: attribute and occupancy query)
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
  }
  if (resident_blocks_per_cu) {
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(resident_blocks_per_cu, fn, (int)threads, lds_bytes);
    if (e != hipSuccess) return e;
  }
  KParams pc = p;
  void* args[] = { &pc, &rays_dev, &n, &out_dev, &head_dev };
  return hipLaunchKernel(fn, dim3(n_blocks), dim3(threads), args, lds_bytes, stream);
}

hipError_t launch_trace_rays(const KParams& p, int kind, const float* rays_dev, uint32_t n, int4* out_dev,
                             hipStream_t stream)
{
  if (n == 0) return hipSuccess;
  if (kind == 3) {   // four-wide walk; p.stack_lds_entries covers the whole stack (3 x depth of the wide tree)
    const size_t lds = (size_t)p.stack_lds_entries * 64u * sizeof(uint2);
    if (lds > 64 * 1024) {
      const void* fn = p.wide8 == 1u ? reinterpret_cast<const void*>(pt_trace_rays_wide_kernel<1>)
                     : (p.wide8 == 2u ? reinterpret_cast<const void*>(pt_trace_rays_wide_kernel<2>) : reinterpret_cast<const void*>(pt_trace_rays_wide_kernel<0>));
      hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
    }
    if (p.wide8 == 1u) hipLaunchKernelGGL(pt_trace_rays_wide_kernel<1>, dim3((n + 63u) / 64u), dim3(64), lds, stream, p, rays_dev, n, out_dev);
    else if (p.wide8 == 2u) hipLaunchKernelGGL(pt_trace_rays_wide_kernel<2>, dim3((n + 63u) / 64u), dim3(64), lds, stream, p, rays_dev, n, out_dev);
    else hipLaunchKernelGGL(pt_trace_rays_wide_kernel<0>, dim3((n + 63u) / 64u), dim3(64), lds, stream, p, rays_dev, n, out_dev);
    return hipGetLastError();
  }
  dim3 grid((n + 255u) / 256u);
  if (kind == 1) hipLaunchKernelGGL(pt_trace_rays_kernel<1>, grid, dim3(256), 0, stream, p, rays_dev, n, out_dev);
  else hipLaunchKernelGGL(pt_trace_rays_kernel<2>, grid, dim3(256), 0, stream, p, rays_dev, n, out_dev);
  return hipGetLastError();
}

#else
} // namespace (ptamd_fma in this translation unit)
// C entry points of the contracted instantiation (pt_kernels_fma.hip): ptamd_api.cpp calls these for PTAMD_KERNEL_BVH_RESTART_FMA
extern "C" hipError_t ptamd_fma_restart_blocks_per_cu(int lds_resident, size_t lds_bytes, int* out)
{
  return ptamd::restart_blocks_per_cu(lds_resident != 0, lds_bytes, out);
}
extern "C" hipError_t ptamd_fma_launch_restart(const void* kparams, int lds_resident, size_t lds_bytes, uint32_t n_blocks, hipStream_t stream)
{
  return ptamd::launch_megakernel_restart(*reinterpret_cast<const ptamd::KParams*>(kparams), lds_resident != 0, lds_bytes, false, n_blocks, stream);
}
namespace ptamd {
#endif
} // namespace ptamd
