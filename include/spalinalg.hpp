// spalinalg.hpp -- header-only C++ mirror of the reference's public matrix
// types over the C ABI of libspal_hip.so (include/spal.h).
//
// The reference is a Rust crate (lokyhark/spalinalg); there is no rustc in the
// build image, so the host side above the C ABI is written in C++ with the
// same names, argument meaning and error behaviour (paths relative to the
// reference root):
//
//   spalinalg::CsrMatrix<T>::CsrMatrix(nrows, ncols, rowptr, colind, values)
//       == CsrMatrix::new, incl. its panics           src/csr.rs:137-164
//   nrows() ncols() rowptr() colind() values() nnz()     src/csr.rs:200-289
//   a * x  (x a dense std::vector<T>)
//       == `&a * &x_as_matrix`, bound in Rust as
//          impl Mul<&[T]> for &CsrMatrix<T>              src/csr/ops/mul.rs:5-59
//   CsrMatrix<T>::from(coo)  == CsrMatrix::from(&coo)    src/csr/conv/coo.rs:3-116
//   CscMatrix<T>, CooMatrix<T> likewise                  src/csc.rs, src/coo.rs
//
// A failed `assert!` in the reference is a panic; here it is a
// spalinalg::Panic exception (the Rust shim in rust_shim/ turns the same
// statuses into panic!()).  Every other failure (no device, HIP error, out of
// memory) is a spalinalg::Error.  There is no CPU fallback.
//
// T is exactly the two `Scalar` impls: float, double (src/scalar.rs:56-57).
#pragma once

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

#include "spal.h"

namespace spalinalg {

using usize = uint64_t;

struct Panic : std::logic_error {
    int status;
    Panic(int st, const std::string &m) : std::logic_error(m), status(st) {}
};
struct Error : std::runtime_error {
    int status;
    Error(int st, const std::string &m) : std::runtime_error(m), status(st) {}
};

namespace detail {
inline void check(int st) {
    if (st == SPAL_OK) return;
    const std::string msg = spal_last_error();
    if (st == SPAL_ERR_INVALID_ARGUMENT || st == SPAL_ERR_INVARIANT || st == SPAL_ERR_INDEX_OUT_OF_BOUNDS)
        throw Panic(st, msg);
    throw Error(st, msg);
}
template <typename T> struct Abi;
template <> struct Abi<double> {
    static constexpr auto csr_create = spal_csr_create_f64;
    static constexpr auto csr_spmv = spal_csr_spmv_f64;
    static constexpr auto csr_download = spal_csr_download_f64;
    static constexpr auto csc_create = spal_csc_create_f64;
    static constexpr auto csc_spmv = spal_csc_spmv_f64;
    static constexpr auto coo_to_csr = spal_coo_to_csr_f64;
    static constexpr auto coo_to_csc = spal_coo_to_csc_f64;
    static constexpr auto csc_download = spal_csc_download_f64;
};
template <> struct Abi<float> {
    static constexpr auto csr_create = spal_csr_create_f32;
    static constexpr auto csr_spmv = spal_csr_spmv_f32;
    static constexpr auto csr_download = spal_csr_download_f32;
    static constexpr auto csc_create = spal_csc_create_f32;
    static constexpr auto csc_spmv = spal_csc_spmv_f32;
    static constexpr auto coo_to_csr = spal_coo_to_csr_f32;
    static constexpr auto coo_to_csc = spal_coo_to_csc_f32;
    static constexpr auto csc_download = spal_csc_download_f32;
};
struct CsrDeleter { void operator()(spal_csr *h) const { spal_csr_destroy(h); } };
struct CscDeleter { void operator()(spal_csc *h) const { spal_csc_destroy(h); } };
}  // namespace detail

template <typename T> class CooMatrix;
template <typename T> class CscMatrix;

// ---------------------------------------------------------------------------
// CsrMatrix<T>                                     reference src/csr.rs:66-72
// ---------------------------------------------------------------------------
template <typename T>
class CsrMatrix {
    static_assert(std::is_same<T, float>::value || std::is_same<T, double>::value,
                  "Scalar is implemented for f32 and f64 only");

  public:
    // CsrMatrix::new -- panics exactly where the reference does.
    CsrMatrix(usize nrows, usize ncols, std::vector<usize> rowptr, std::vector<usize> colind,
              std::vector<T> values)
        : nrows_(nrows), ncols_(ncols), rowptr_(std::move(rowptr)), colind_(std::move(colind)),
          values_(std::move(values)) {
        int reason = 0;
        detail::check(spal_csr_validate(nrows_, ncols_, rowptr_.data(), rowptr_.size(), colind_.data(),
                                        colind_.size(), values_.size(), &reason));
    }

    usize nrows() const { return nrows_; }
    usize ncols() const { return ncols_; }
    const std::vector<usize> &rowptr() const { return rowptr_; }
    const std::vector<usize> &colind() const { return colind_; }
    const std::vector<T> &values() const { return values_; }
    usize nnz() const { return rowptr_[nrows_]; }

    // Device copy (created on first use, owned by this object).
    spal_csr_t device_handle(int device = 0) const {
        if (!dev_) {
            spal_csr_t h = nullptr;
            detail::check(detail::Abi<T>::csr_create(device, nrows_, ncols_, rowptr_.data(), rowptr_.size(),
                                                     colind_.data(), colind_.size(), values_.data(),
                                                     values_.size(), &h));
            dev_.reset(h);
        }
        return dev_.get();
    }

    // y = A * x.  Panics when x.len() != ncols (assert_eq!, src/csr/ops/mul.rs:9).
    std::vector<T> operator*(const std::vector<T> &x) const {
        if (x.size() != ncols_)
            throw Panic(SPAL_ERR_INVALID_ARGUMENT, "assertion failed: `(left == right)` ncols vs x.len()");
        std::vector<T> y(nrows_);
        detail::check(detail::Abi<T>::csr_spmv(device_handle(), x.data(), x.size(), y.data(), y.size()));
        return y;
    }

    // CsrMatrix::from(&coo): assembled on the device, bit-identical to the reference.
    static CsrMatrix from(const CooMatrix<T> &coo, int device = 0);
    // CsrMatrix::from(&csc)  (src/csr/conv/csc.rs:4-52): device stable sort by row.
    static CsrMatrix from(const CscMatrix<T> &csc, int device = 0);

  private:
    friend class CscMatrix<T>;
    struct Trusted {};
    CsrMatrix(Trusted, usize nrows, usize ncols, std::vector<usize> rp, std::vector<usize> ci,
              std::vector<T> va)
        : nrows_(nrows), ncols_(ncols), rowptr_(std::move(rp)), colind_(std::move(ci)),
          values_(std::move(va)) {}
    static CsrMatrix adopt(spal_csr_t h) {   // downloads h into a host matrix that also owns h
        std::unique_ptr<spal_csr, detail::CsrDeleter> guard(h);
        uint64_t nr = 0, nc = 0, nz = 0;
        int es = 0;
        detail::check(spal_csr_shape(h, &nr, &nc, &nz, &es));
        std::vector<usize> rp(nr + 1), ci(nz);
        std::vector<T> va(nz);
        detail::check(detail::Abi<T>::csr_download(h, rp.data(), ci.data(), va.data()));
        // struct-literal construction like the reference (src/csr/conv/coo.rs:108-114)
        CsrMatrix out(Trusted{}, nr, nc, std::move(rp), std::move(ci), std::move(va));
        out.dev_ = std::move(guard);
        return out;
    }
    usize nrows_, ncols_;
    std::vector<usize> rowptr_, colind_;
    std::vector<T> values_;
    mutable std::unique_ptr<spal_csr, detail::CsrDeleter> dev_;
};

// ---------------------------------------------------------------------------
// CscMatrix<T>                                     reference src/csc.rs:66-72
// ---------------------------------------------------------------------------
template <typename T>
class CscMatrix {
  public:
    CscMatrix(usize nrows, usize ncols, std::vector<usize> colptr, std::vector<usize> rowind,
              std::vector<T> values)
        : nrows_(nrows), ncols_(ncols), colptr_(std::move(colptr)), rowind_(std::move(rowind)),
          values_(std::move(values)) {
        int reason = 0;
        detail::check(spal_csc_validate(nrows_, ncols_, colptr_.data(), colptr_.size(), rowind_.data(),
                                        rowind_.size(), values_.size(), &reason));
    }
    usize nrows() const { return nrows_; }
    usize ncols() const { return ncols_; }
    const std::vector<usize> &colptr() const { return colptr_; }
    const std::vector<usize> &rowind() const { return rowind_; }
    const std::vector<T> &values() const { return values_; }
    usize nnz() const { return colptr_[ncols_]; }

    spal_csc_t device_handle(int device = 0) const {
        if (!dev_) {
            spal_csc_t h = nullptr;
            detail::check(detail::Abi<T>::csc_create(device, nrows_, ncols_, colptr_.data(), colptr_.size(),
                                                     rowind_.data(), rowind_.size(), values_.data(),
                                                     values_.size(), &h));
            dev_.reset(h);
        }
        return dev_.get();
    }
    // y = A * x; panics when x.len() != ncols (src/csc/ops/mul.rs:9).
    std::vector<T> operator*(const std::vector<T> &x) const {
        if (x.size() != ncols_)
            throw Panic(SPAL_ERR_INVALID_ARGUMENT, "assertion failed: `(left == right)` ncols vs x.len()");
        std::vector<T> y(nrows_);
        detail::check(detail::Abi<T>::csc_spmv(device_handle(), x.data(), x.size(), y.data(), y.size()));
        return y;
    }
    // CscMatrix::from(&csr)  (src/csc/conv/csr.rs:4-52) and CscMatrix::from(&coo)
    // (src/csc/conv/coo.rs:3-116), both on the device.
    static CscMatrix from(const CsrMatrix<T> &csr, int device = 0) {
        spal_csc_t h = nullptr;
        detail::check(spal_csr_to_csc(csr.device_handle(device), &h));
        return adopt(h);
    }
    static CscMatrix from(const CooMatrix<T> &coo, int device = 0);

  private:
    friend class CsrMatrix<T>;
    struct Trusted {};
    CscMatrix(Trusted, usize nrows, usize ncols, std::vector<usize> cp, std::vector<usize> ri,
              std::vector<T> va)
        : nrows_(nrows), ncols_(ncols), colptr_(std::move(cp)), rowind_(std::move(ri)),
          values_(std::move(va)) {}
    static CscMatrix adopt(spal_csc_t h) {
        std::unique_ptr<spal_csc, detail::CscDeleter> guard(h);
        uint64_t nr = 0, nc = 0, nz = 0;
        int es = 0;
        detail::check(spal_csc_shape(h, &nr, &nc, &nz, &es));
        std::vector<usize> cp(nc + 1), ri(nz);
        std::vector<T> va(nz);
        detail::check(detail::Abi<T>::csc_download(h, cp.data(), ri.data(), va.data()));
        CscMatrix out(Trusted{}, nr, nc, std::move(cp), std::move(ri), std::move(va));
        out.dev_ = std::move(guard);
        return out;
    }
    usize nrows_, ncols_;
    std::vector<usize> colptr_, rowind_;
    std::vector<T> values_;
    mutable std::unique_ptr<spal_csc, detail::CscDeleter> dev_;
};

// ---------------------------------------------------------------------------
// CooMatrix<T>                                      reference src/coo.rs:53-57
// Insertion order is significant: the conversion sums duplicates in it.
// ---------------------------------------------------------------------------
template <typename T>
class CooMatrix {
  public:
    // CooMatrix::new                                         src/coo.rs:104-112
    CooMatrix(usize nrows, usize ncols) : nrows_(nrows), ncols_(ncols) {
        if (!(nrows > 0)) throw Panic(SPAL_ERR_INVARIANT, "assertion failed: nrows > 0");
        if (!(ncols > 0)) throw Panic(SPAL_ERR_INVARIANT, "assertion failed: ncols > 0");
    }
    static CooMatrix with_capacity(usize nrows, usize ncols, usize capacity) {
        CooMatrix m(nrows, ncols);
        m.rows_.reserve(capacity); m.cols_.reserve(capacity); m.vals_.reserve(capacity);
        return m;
    }
    // CooMatrix::with_triplets                               src/coo.rs:260-288
    static CooMatrix with_triplets(usize nrows, usize ncols, const std::vector<usize> &rowind,
                                   const std::vector<usize> &colind, const std::vector<T> &values) {
        CooMatrix m(nrows, ncols);
        if (rowind.size() != values.size()) throw Panic(SPAL_ERR_INVARIANT, "assertion failed: rowind.len() == values.len()");
        if (colind.size() != values.size()) throw Panic(SPAL_ERR_INVARIANT, "assertion failed: colind.len() == values.len()");
        for (usize r : rowind) if (!(r < nrows)) throw Panic(SPAL_ERR_INDEX_OUT_OF_BOUNDS, "assertion failed: *row < nrows");
        for (usize c : colind) if (!(c < ncols)) throw Panic(SPAL_ERR_INDEX_OUT_OF_BOUNDS, "assertion failed: *col < ncols");
        m.rows_ = rowind; m.cols_ = colind; m.vals_ = values;
        return m;
    }
    // CooMatrix::with_entries                                src/coo.rs:204-221
    static CooMatrix with_entries(usize nrows, usize ncols, const std::vector<std::tuple<usize, usize, T>> &entries) {
        CooMatrix m(nrows, ncols);
        for (const auto &e : entries) m.push(std::get<0>(e), std::get<1>(e), std::get<2>(e));
        return m;
    }
    // CooMatrix::push                                        src/coo.rs:431-435
    void push(usize row, usize col, T value) {
        if (!(row < nrows_)) throw Panic(SPAL_ERR_INDEX_OUT_OF_BOUNDS, "assertion failed: row < self.nrows");
        if (!(col < ncols_)) throw Panic(SPAL_ERR_INDEX_OUT_OF_BOUNDS, "assertion failed: col < self.ncols");
        rows_.push_back(row); cols_.push_back(col); vals_.push_back(value);
    }
    usize nrows() const { return nrows_; }
    usize ncols() const { return ncols_; }
    usize length() const { return vals_.size(); }
    // the three components of `iter()` (src/coo.rs:491-495), unzipped
    const std::vector<usize> &rows() const { return rows_; }
    const std::vector<usize> &cols() const { return cols_; }
    const std::vector<T> &vals() const { return vals_; }

  private:
    usize nrows_, ncols_;
    std::vector<usize> rows_, cols_;
    std::vector<T> vals_;
};

template <typename T>
CsrMatrix<T> CsrMatrix<T>::from(const CooMatrix<T> &coo, int device) {
    spal_csr_t h = nullptr;
    detail::check(detail::Abi<T>::coo_to_csr(device, coo.nrows(), coo.ncols(), coo.length(),
                                             coo.rows().data(), coo.cols().data(), coo.vals().data(), &h));
    return adopt(h);
}

template <typename T>
CsrMatrix<T> CsrMatrix<T>::from(const CscMatrix<T> &csc, int device) {
    spal_csr_t h = nullptr;
    detail::check(spal_csc_to_csr(csc.device_handle(device), &h));
    return adopt(h);
}

template <typename T>
CscMatrix<T> CscMatrix<T>::from(const CooMatrix<T> &coo, int device) {
    spal_csc_t h = nullptr;
    detail::check(detail::Abi<T>::coo_to_csc(device, coo.nrows(), coo.ncols(), coo.length(),
                                             coo.rows().data(), coo.cols().data(), coo.vals().data(), &h));
    return adopt(h);
}

}  // namespace spalinalg
