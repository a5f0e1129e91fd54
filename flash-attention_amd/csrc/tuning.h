// Tuning knobs of the launchers.
//
// The shipped library (libmyfa_gfx950.so) has none: tuning() returns compile-time defaults and
// reads nothing from the process environment.  The -DMYFA_TUNING flavour
// (libmyfa_gfx950_tuning.so, used by tools/ and by the tests that compare kernel variants)
// keeps one mutable record per process and exports two extra C symbols to set it,
// myfa_tuning_set(key, value) and myfa_tuning_set_ptr(key, pointer); both flavours export the
// same ABI otherwise.
#ifndef MYFA_TUNING_H_
#define MYFA_TUNING_H_

namespace myfa {

struct Tuning {
    // attention forward, f16/bf16:
    //   0 = 32x32x16 MFMA, one barrier per tile, register-staged K/V
    //   1 = 32x32x16 MFMA, ping-pong of the two waves of a SIMD, LDS-DMA rings
    //   2 = 16x16x32 MFMA, same schedule as 1
    //   3 = 16x16x32 MFMA, resident workgroups that stream through a list of query blocks
    //  -1 = the library's choice per shape
    int attn_variant = -1;
    // deferred-max threshold in log2 units (0 = rescale whenever a row max grows: the textbook
    // recurrence); < 0 = the library's choice per kernel and element type
    float attn_thr = -1.f;
    // 16x16x32 build: 1 = weights formed against the standing max, the max looked for only when a row
    // sum passes 2^attn_thr; 0 = the max of every tile, vote on its growth; -1 = the library's choice;
    // 2 (resident-workgroup build, head dims above 64, no mask) = as 1 with the row sums formed by the
    // matrix cores from the rounded weights and the go-back test on the largest weight
    int attn_spec = -1;
    // timing probes of the stamped 16x16x32 build (wrong results): 8 = no tile fetches in the loop,
    // 16 = no vector phase
    int attn_xflags = 0;
    // resident-workgroup build, speculative softmax: 1 = Q pre-multiplied by scale * log2(e) and the
    // running max folded into the accumulators' start value, 0 = not, -1 = the library's choice
    int attn_pre = -1;
    // resident-workgroup build, head dims up to 64: 1 = one workgroup per CU (256 registers per lane),
    // else two (128)
    int attn_occ = 0;
    // resident-workgroup build: workgroups in the grid (0 = as many as the chip holds at once)
    int attn_wgs = 0;
    // device buffer for the stamped (cycle-counting) diagnostic build, 24 u64 per workgroup
    unsigned long long* attn_dbg = nullptr;
    // float32 attention, keys split over workgroups: workgroup count the split aims at (0 = the default, 256)
    int attn_f32_wgs = 0;
    // row softmax launch shape override: threads per row, items per thread, map, nt, pad (tpr = 0: none)
    int softmax_shape[5] = {0, 0, 0, 3, 0};
    // column softmax with the rows of a column block split over workgroups: workgroup count the split
    // aims at (0 = the default, 1024; -1 = never split, as without a workspace)
    int softmax_split_wgs = 0;
    // column softmax, 1024-thread register-resident kernel: 0 = library's choice, -1 = never
    int softmax_cols_big = 0;
    // float32 matmul block edge: 0 = library's choice, 64 or 128
    int matmul_block = 0;
    // float32 matmul, 64 x 64 kernel: 0 = library's choice, 1 = k never split, 2 = k split over two workgroups
    int matmul_ksplit = 0;
};

#ifdef MYFA_TUNING
Tuning& tuning();
#else
inline const Tuning& tuning() {
    static const Tuning t{};
    return t;
}
#endif

}  // namespace myfa
#endif  // MYFA_TUNING_H_
