// The split-f16 output GEMM with the RQ-spline epilogue for the spline layouts other than the plain / circular one:
// identity boundary slopes (spline.py:359-380) and learnable bounds (spline.py:384-410) change the number of
// parameters per feature P, and with it the column tile (16 features x P parameters) of the kernel.  A translation
// unit of its own: every layout is one more instantiation of the 512-register GEMM kernel (split_gemm_kernel.h), and
// the build compiles the .hip files in parallel.
#include "split_gemm_kernel.h"

namespace tfep {

int launch_split_fused_layouts(const GemmArgs& g, int n_rows_w, int K, int P, int n_col_tiles, hipStream_t s) {
    if (P == 3 * K + 1) {
        // identity slopes with both bounds learnable: the parameter count of the plain layout, an epilogue of its own
        const SplineFlags& f = g.fu.sf;
        TFEP_REQUIRE(f.identity && f.learn_lower && f.learn_upper && !f.circular, "fused split: not a layout of this file");
        if (K == 8) return launch_split<25, EPI_SPLINE_IDB, 25, 8>(g, n_rows_w, n_col_tiles, s);
        if (K == 5) return launch_split<16, EPI_SPLINE_IDB, 16, 5>(g, n_rows_w, n_col_tiles, s);
        if (K == 4) return launch_split<13, EPI_SPLINE_IDB, 13, 4>(g, n_rows_w, n_col_tiles, s);
        return fail(TFEP_ERR_UNSUPPORTED, "fused split: no kernel for %d bins", K);
    }
#define TFEP_SPLIT_SPLINE(KK, PP) \
    if (K == KK && P == PP) return launch_split<PP, EPI_SPLINE, PP, KK>(g, n_rows_w, n_col_tiles, s);
    // 8 bins: identity slopes (plain: 23; circular or with one learnable bound: 24); learnable bounds without identity
    // slopes: 26 / 27 accumulator tiles -- 416 / 432 accumulator registers; the k-loop of the 27-tile kernel reloads three
    // spilled address registers per k-tile, no accumulator leaves the register file (tools/scan_hazards.sh)
    TFEP_SPLIT_SPLINE(8, 23) TFEP_SPLIT_SPLINE(8, 24) TFEP_SPLIT_SPLINE(8, 26) TFEP_SPLIT_SPLINE(8, 27)
    TFEP_SPLIT_SPLINE(5, 14) TFEP_SPLIT_SPLINE(5, 15) TFEP_SPLIT_SPLINE(5, 17) TFEP_SPLIT_SPLINE(5, 18)
    TFEP_SPLIT_SPLINE(4, 11) TFEP_SPLIT_SPLINE(4, 12) TFEP_SPLIT_SPLINE(4, 14) TFEP_SPLIT_SPLINE(4, 15)
#undef TFEP_SPLIT_SPLINE
    return fail(TFEP_ERR_UNSUPPORTED, "fused split: no kernel for a %d-bin spline of %d parameters per feature", K, P);
}

}  // namespace tfep
