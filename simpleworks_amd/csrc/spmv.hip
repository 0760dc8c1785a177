// spmv.hip — K3: R1CS sparse mat-vec over BLS12-377 Fr for gfx950.
//
// Replaces the row-wise inner products of ark-marlin's AHPForR1CS::prover_init (z_A = A z, z_B = B z),
// the column-wise accumulation of calculate_t (through the transposed matrices) and ConstraintSystem::is_satisfied
// (/root/reference/src/merkle_tree/simple_merkle_tree.rs:197-199).
//
// R1CS matrices are extremely ragged: gadget rows hold 0-3 non-zeros (src/gadgets/uint8.rs:117-118) while the
// transposed matrices have a few rows with ~|H| entries (the constant-one column, a variable used by every row).
// A row-per-lane CSR kernel serialises on those rows (measured: 1.4 s of a 1.57 s prove at 2^20), so the kernel is
// non-zero-parallel instead, using the fact that (Fr, +) is a group:
//     1. prod[k] = val[k] * z[col[k]]                     one lane per non-zero, coalesced val/col, gathered z
//     2. T = suffix sums of prod                           blocked two-pass scan (devops.cuh suffix_recurrence)
//     3. out[r] = T[rowptr[r]] - T[rowptr[r+1]]            one lane per row
// Every step is balanced whatever the row-length distribution.  HBM-bound integer work, no LDS, no MFMA.
// Algorithmic bytes: 68 B per non-zero + 36 B per row.
#include "devops.cuh"

namespace swm {

int spmv_run(swm_ctx* ctx, const void* d_rowptr, const void* d_col, const void* d_val, const void* d_z, void* d_out,
             size_t rows) {
    if (rows == 0) return SWM_OK;
    try {
        ctx->stat_spmv_calls++;
        ctx->stat_spmv_rows += rows;
        const uint32_t* rowptr = (const uint32_t*)d_rowptr;
        const uint32_t* col = (const uint32_t*)d_col;
        const Fr* val = (const Fr*)d_val;
        const Fr* z = (const Fr*)d_z;
        Fr* out = (Fr*)d_out;
        uint32_t nnz = 0;
        hip_check(ctx, hipMemcpyAsync(&nnz, rowptr + rows, 4, hipMemcpyDeviceToHost, ctx->stream), "d2h");
        hip_check(ctx, hipStreamSynchronize(ctx->stream), "sync");
        DVec prod(ctx, (size_t)nnz + 1);
        Fr* pp = prod.p;
        ew(ctx, "spmv_products", (size_t)nnz + 1, [=] __device__(size_t k) {
            if (k < nnz) {
                Fr c = val[k];
                Fr zv = z[col[k]];
                pp[k] = fp_is_one(c) ? zv : fp_mul(zv, c);  // coeff.is_one() shortcut as in prover_init
            } else {
                pp[k] = fp_zero<Fr>();  // T[nnz] = 0
            }
        });
        suffix_recurrence(ctx, pp, (size_t)nnz + 1, 1, fp_one<Fr>());
        ew(ctx, "spmv_rows", rows, [=] __device__(size_t r) { out[r] = fp_sub(pp[rowptr[r]], pp[rowptr[r + 1]]); });
        return SWM_OK;
    } catch (const MarlinError& e) {
        return set_err(ctx, e.code, "%s", e.what());
    }
}

}  // namespace swm
