// blitzdg-mi355x: basic value/array types for the host side.
//
// Mirrors the vocabulary of the reference's include/Types.hpp:16-31
// (real_type = double, index_type = int, real_matrix_type row-major with the
// LAST index contiguous, index_hashmap) without blitz++: arrays here are plain
// contiguous std::vector storage with (i) / (i,j) accessors, so tables can be
// handed to the C-ABI / HIP side as raw pointers with no copies.
#pragma once
#include <cstddef>
#include <memory>
#include <new>
#include <stdexcept>
#include <type_traits>
#include <utility>
#include <unordered_map>
#include <vector>

namespace blitzdg {

using real_type = double;
using index_type = int;

namespace detail {
/// std::allocator whose value-less construct() default-initialises (no zero fill), so that
/// resizeUninitialized() below can leave first touch of a large table to the parallel loop that fills it.
template <typename T>
struct default_init_allocator : std::allocator<T> {
    template <typename U> struct rebind { using other = default_init_allocator<U>; };
    using std::allocator<T>::allocator;
    template <typename U> void construct(U* p) noexcept(std::is_nothrow_default_constructible<U>::value) {
        ::new (static_cast<void*>(p)) U;
    }
    template <typename U, typename... Args> void construct(U* p, Args&&... args) {
        ::new (static_cast<void*>(p)) U(std::forward<Args>(args)...);
    }
};
template <typename T> using storage = std::vector<T, default_init_allocator<T>>;
} // namespace detail

/// Dense 1-D array. `length(0)` / `size()` follow the reference's blitz spelling.
template <typename T>
class vector_type {
public:
    vector_type() = default;
    explicit vector_type(index_type n) : d_(static_cast<std::size_t>(n), T{}) {}
    vector_type(index_type n, T fill) : d_(static_cast<std::size_t>(n), fill) {}
    T& operator()(index_type i) { return d_[static_cast<std::size_t>(i)]; }
    const T& operator()(index_type i) const { return d_[static_cast<std::size_t>(i)]; }
    T& operator[](index_type i) { return d_[static_cast<std::size_t>(i)]; }
    const T& operator[](index_type i) const { return d_[static_cast<std::size_t>(i)]; }
    index_type length(int = 0) const { return static_cast<index_type>(d_.size()); }
    index_type size() const { return static_cast<index_type>(d_.size()); }
    index_type numElements() const { return size(); }
    T* data() { return d_.data(); }
    const T* data() const { return d_.data(); }
    void resize(index_type n) { d_.assign(static_cast<std::size_t>(n), T{}); }
    /// Contents unspecified: for tables every entry of which is written next.
    void resizeUninitialized(index_type n) { d_.clear(); d_.resize(static_cast<std::size_t>(n)); }
    void fill(T v) { d_.assign(d_.size(), v); }
    typename detail::storage<T>::iterator begin() { return d_.begin(); }
    typename detail::storage<T>::iterator end() { return d_.end(); }
    typename detail::storage<T>::const_iterator begin() const { return d_.begin(); }
    typename detail::storage<T>::const_iterator end() const { return d_.end(); }
private:
    detail::storage<T> d_;
};

/// Dense 2-D array, row-major: element (i,j) at data()[i*cols()+j].
/// For every (rows, K) field/table this makes the element index K contiguous,
/// which is also the device layout (one wavefront lane per element).
template <typename T>
class matrix_type {
public:
    matrix_type() = default;
    matrix_type(index_type r, index_type c)
        : r_(r), c_(c), d_(static_cast<std::size_t>(r) * static_cast<std::size_t>(c), T{}) {}
    T& operator()(index_type i, index_type j) {
        return d_[static_cast<std::size_t>(i) * c_ + static_cast<std::size_t>(j)];
    }
    const T& operator()(index_type i, index_type j) const {
        return d_[static_cast<std::size_t>(i) * c_ + static_cast<std::size_t>(j)];
    }
    index_type rows() const { return r_; }
    index_type cols() const { return c_; }
    index_type length(int dim) const { return dim == 0 ? r_ : c_; }
    std::size_t numElements() const { return d_.size(); }
    T* data() { return d_.data(); }
    const T* data() const { return d_.data(); }
    void resize(index_type r, index_type c) {
        r_ = r; c_ = c;
        d_.assign(static_cast<std::size_t>(r) * static_cast<std::size_t>(c), T{});
    }
    /// Contents unspecified: for tables every entry of which is written next.
    void resizeUninitialized(index_type r, index_type c) {
        r_ = r; c_ = c;
        d_.clear();
        d_.resize(static_cast<std::size_t>(r) * static_cast<std::size_t>(c));
    }
    void fill(T v) { d_.assign(d_.size(), v); }
private:
    index_type r_ = 0, c_ = 0;
    detail::storage<T> d_;
};

using real_vector_type = vector_type<real_type>;
using index_vector_type = vector_type<index_type>;
using real_matrix_type = matrix_type<real_type>;
using index_matrix_type = matrix_type<index_type>;
using index_hashmap = std::unordered_map<index_type, std::vector<index_type>>;

} // namespace blitzdg
