// pk_codec.h — the part of the proving-key codec (ark-serialize layout of ark-marlin's IndexProverKey) that needs no
// GPU: the verifying key, the index info, the three matrices and the SHAPES of the derived tables in front of the committer
// key.  Shared by marlin.hip (pk_deserialize continues with the committer key, whose points are decompressed and checked on
// the device) and by the sanitizer harness tests/native/host_fuzz.cpp.  Every malformed input is a MarlinError.
#pragma once
#include <algorithm>
#include "ahp.h"

namespace swm {

struct HostCsr {
    std::vector<uint32_t> rowptr, col;
    std::vector<Fr> val;
    size_t rows() const { return rowptr.size() - 1; }
    size_t nnz() const { return col.size(); }
};
inline HostCsr get_matrix(ByteReader& r, uint64_t ncols) {
    HostCsr m;
    uint64_t rows = r.u64();
    if (rows > (1ull << 31)) throw MarlinError(SWM_ERR_SERIALIZATION, "bad matrix header");
    m.rowptr.assign(1, 0);
    for (uint64_t i = 0; i < rows; i++) {
        uint64_t len = r.u64();
        if (len > r.n - r.pos) throw MarlinError(SWM_ERR_SERIALIZATION, "bad matrix row");
        for (uint64_t k = 0; k < len; k++) {
            m.val.push_back(r.fr());
            uint64_t c = r.u64();
            if (c >= ncols) throw MarlinError(SWM_ERR_SERIALIZATION, "matrix column out of range");
            m.col.push_back((uint32_t)c);
        }
        m.rowptr.push_back((uint32_t)m.col.size());
    }
    return m;
}
inline void skip_fr_vec(ByteReader& r, uint64_t expect_max) {
    uint64_t n = r.u64();
    if (n > expect_max) throw MarlinError(SWM_ERR_SERIALIZATION, "vector longer than its domain");
    r.take(n * 32);
}
inline void skip_domain(ByteReader& r, uint64_t size) {
    if (*r.take(1) != 0) throw MarlinError(SWM_ERR_SERIALIZATION, "not a radix-2 domain");
    if (r.u64() != size) throw MarlinError(SWM_ERR_SERIALIZATION, "domain size does not match the index");
    r.take(4 + 5 * 32);
}
struct PkPrefix {
    VerifyingKey vk;
    IndexInfo info;
    HostCsr a, b, c;
    uint64_t H = 0, K = 0, X = 0, B = 0;
    unsigned logH = 0, logK = 0, logX = 0, logB = 0;
};
// reads up to (not including) the committer key; r.pos is left at its first byte
inline PkPrefix pk_parse_prefix(ByteReader& r) {
    PkPrefix pre;
    pre.vk = read_verifying_key(r);
    uint64_t nr = r.u64();
    if (nr != pre.vk.index_comms.size()) throw MarlinError(SWM_ERR_SERIALIZATION, "index_comm_rands does not match index_comms");
    for (uint64_t i = 0; i < nr; i++) {
        skip_fr_vec(r, 4);
        if (r.boolean()) skip_fr_vec(r, 4);
    }
    pre.info.num_variables = r.u64();
    pre.info.num_constraints = r.u64();
    pre.info.num_non_zero = r.u64();
    pre.info.num_instance_variables = r.u64();
    if (pre.info.num_variables != pre.vk.info.num_variables || pre.info.num_constraints != pre.vk.info.num_constraints ||
        pre.info.num_non_zero != pre.vk.info.num_non_zero || pre.info.num_instance_variables != pre.vk.info.num_instance_variables ||
        pre.info.num_constraints != pre.info.num_variables || pre.info.num_variables > (1ull << 30) ||
        pre.info.num_non_zero > (1ull << 31) || pre.info.num_non_zero == 0)
        throw MarlinError(SWM_ERR_SERIALIZATION, "index info is inconsistent");
    pre.a = get_matrix(r, pre.info.num_variables);
    pre.b = get_matrix(r, pre.info.num_variables);
    pre.c = get_matrix(r, pre.info.num_variables);
    if (pre.a.rows() != pre.info.num_constraints || pre.b.rows() != pre.info.num_constraints ||
        pre.c.rows() != pre.info.num_constraints)
        throw MarlinError(SWM_ERR_SERIALIZATION, "matrix shape does not match index info");
    if (std::max(pre.a.nnz(), std::max(pre.b.nnz(), pre.c.nnz())) > pre.info.num_non_zero)
        throw MarlinError(SWM_ERR_SERIALIZATION, "matrix density exceeds index info");
    HDomain dh(pre.info.num_constraints), dk(pre.info.num_non_zero), dx(pre.info.num_instance_variables);
    HDomain db(3 * dk.size - 3);
    pre.H = dh.size; pre.logH = dh.log;
    pre.K = dk.size; pre.logK = dk.log;
    pre.X = dx.size; pre.logX = dx.log;
    pre.B = db.size; pre.logB = db.log;
    if (pre.X >= pre.H) throw MarlinError(SWM_ERR_SERIALIZATION, "index without witness variables");
    for (int m = 0; m < 3; m++) {  // derived data: shape-checked, then recomputed below
        for (int j = 0; j < 4; j++) {
            uint64_t ll = r.u64();
            if (ll > 64) throw MarlinError(SWM_ERR_SERIALIZATION, "bad polynomial label");
            r.take(ll);
            skip_fr_vec(r, pre.K);
            if (r.boolean()) r.u64();
            if (r.boolean()) r.u64();
        }
        for (int j = 0; j < 3; j++) {
            skip_fr_vec(r, pre.K);
            skip_domain(r, pre.K);
        }
        for (int j = 0; j < 4; j++) {
            skip_fr_vec(r, pre.B);
            skip_domain(r, pre.B);
        }
    }
    return pre;
}

}  // namespace swm
