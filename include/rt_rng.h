/* rt_rng.h — the random-number CONTRACT shared by the device path and the
 * CPU oracle.  This header holds the addressing scheme only (constants and
 * prose); each side carries its own implementation of Philox4x32 so the
 * oracle stays an independent check.
 *
 * Why a contract at all: the reference draws everything from
 * `rand::thread_rng()` (racer-tracer/src/util.rs:9-23), which is OS-seeded
 * and per-thread, so it has no reproducible stream.  We keep the reference's
 * draw ORDER and distribution but address every draw by
 * what it is for instead of by "the n-th call on this thread".  A draw is
 *
 *     out  = philox4x32_R(ctr = {pixel, sample, (segment << 8) | purpose, block},     R = RT_PHILOX_ROUNDS
 *                          key = {seed & 0xffffffff, seed >> 32})
 *     d0   = u53(out[0], out[1])        d1 = u53(out[2], out[3])
 *     u53(hi, lo) = (((uint64)hi << 32 | lo) >> 11) * 2^-53      in [0,1)
 *
 * and `random_double_range(a,b)` (util.rs:14-17) is  a + (b - a) * d.
 *
 * One purpose packs THREE draws into a block, at 42 bits each (RT_RNG_SCATTER below):
 *
 *     e_j  = u42(out[j], out[3] >> 10 j)      j = 0, 1, 2
 *     u42(w, t) = ((uint64)w << 10 | (t & 0x3FF)) * 2^-42         in [0,1)
 *
 * A candidate of random_in_unit_sphere is three coordinates, and the rejection loop's candidates are
 * the largest single consumer of random bits on the path (two blocks per candidate were a quarter of
 * the cornell_box frame's instructions).  The reference's own `gen_range(-1.0..1.0)` (rand 0.8
 * UniformFloat: 52 random mantissa bits, scaled) puts a coordinate on a grid of 2^-51; e_j puts it on
 * a grid of 2^-41.  Nothing the path computes from a candidate can tell the two apart: the acceptance
 * test |p|^2 < 1 changes its answer for a fraction ~2^-40 of candidates, a direction moves by at most
 * 2^-41.  Every other draw keeps 53 bits.
 * Because nothing depends on how many draws came before, a lane never
 * carries generator state, dead draws (lens disk with aperture 0, Metal with
 * fuzz 0, ray time with no moving geometry) can be skipped on the device
 * without changing any other value, and rejection-loop candidates can be
 * evaluated out of order.
 *
 *   pixel   = y * width + x          (global, so results do not depend on
 *                                     tiling or on the number of GPUs)
 *   sample  = sample index s in [0, samples); RT_RNG_SAMPLE_PIXEL for the
 *             one draw a pixel shares between its samples
 *   segment = 0 for the camera ray and the shading of its hit, +1 per bounce
 *             (== max_depth - depth of renderer.rs:41-90)
 *   block   = iteration of a rejection loop (see each purpose)
 */
#ifndef RT_RNG_H
#define RT_RNG_H

#define RT_RNG_SAMPLE_PIXEL 0xFFFFFFFFu /* `sample` of per-pixel draws      */
#define RT_RNG_SAMPLE_TABLE 0xFFFFFFFEu /* `sample` of host-side table draws */

/* purpose                         draws                       reference call site */
#define RT_RNG_PIXEL 0      /* block 0: d0 = ju               cpu.rs:35-36  (once per pixel) */
#define RT_RNG_CAMERA 1     /* block 0: d0 = jv, d1 = time    cpu.rs:39-40, camera.rs:335    */
#define RT_RNG_LENS 2       /* block i: x = R(-1,1)(d0),      util.rs:25-39 iteration i      */
                            /*          y = R(-1,1)(d1)                                      */
#define RT_RNG_SCATTER 3    /* iteration i of random_in_unit_sphere (vec3.rs:424-430):       */
                            /*   block i: x = R(-1,1)(e0), y = R(-1,1)(e1), z = R(-1,1)(e2)  */
                            /* used by Lambertian (lambertian.rs:27) and Metal (metal.rs:34) */
#define RT_RNG_DIELECTRIC 4 /* block 0: d0                    dialectric.rs:44               */
#define RT_RNG_PERLIN 5     /* host: pixel = gradient index, sample = RT_RNG_SAMPLE_TABLE,   */
                            /* segment = perlin index; block 0: x,y  block 1: z              */
                            /* (noise.rs:45-47: Vec3::random_range(-1,1).unit_vector())      */
#define RT_RNG_BVH 6        /* oracle only: pixel = build-node index; axis = d0 < 0.5 ? 0 : 1 */
                            /* (bvh_node.rs:32: random_int_range(0,2))                        */

#define RT_RNG_SCENE 7      /* host: the procedural `random` scene (scene/random.rs:39-70):   */
                            /* pixel = n for the loader's n-th random_double(), sample =      */
                            /* RT_RNG_SAMPLE_TABLE, segment 0, block 0: d0                     */

/* Philox4x32 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3",
 * SC'11) with SEVEN rounds: the paper's Crush-resistant member of the family (Philox4x32-7
 * passes TestU01 BigCrush; the usual -10 adds three rounds of safety margin that a Monte
 * Carlo integrator has no use for, and the generator is ~1/4 of this path's instructions).
 * Both round counts are pinned by Random123's published known-answer vectors
 * (tests/test_oracle_reference_vectors.py).  The reference itself draws from rand 0.8's
 * ThreadRng (ChaCha12), a different family altogether: only the distribution is matched. */
#define RT_PHILOX_ROUNDS 7
#define RT_PHILOX_M0 0xD2511F53u
#define RT_PHILOX_M1 0xCD9E8D57u
#define RT_PHILOX_W0 0x9E3779B9u
#define RT_PHILOX_W1 0xBB67AE85u

#endif /* RT_RNG_H */
