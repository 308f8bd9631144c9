// Gather / reshape vocabulary of the RHS evaluators.
// Same names and semantics as the reference's include/BlitzHelpers.hpp:215-297
// (reshapeMatTo1D, reshape1DToMat, fullToVector, vectorToFull, applyIndexMap):
// "byRows = false" flattens a (rows, cols) matrix column-wise, i.e. entry (i, j)
// goes to position i + rows*j -- the numbering vmapM/vmapP index into.
#pragma once
#include "Types.hpp"
#include <iterator>
#include <type_traits>

namespace blitzdg {

template <typename T>
void fullToVector(const matrix_type<T>& mat, vector_type<T>& vec, bool byRows = true) {
    const index_type R = mat.rows(), C = mat.cols();
    if (byRows) {
        for (index_type i = 0; i < R; ++i)
            for (index_type j = 0; j < C; ++j) vec(i * C + j) = mat(i, j);
    } else {
        for (index_type j = 0; j < C; ++j)
            for (index_type i = 0; i < R; ++i) vec(i + R * j) = mat(i, j);
    }
}

template <typename T>
void vectorToFull(const vector_type<T>& vec, matrix_type<T>& mat, bool byRows = true) {
    const index_type R = mat.rows(), C = mat.cols();
    if (byRows) {
        for (index_type i = 0; i < R; ++i)
            for (index_type j = 0; j < C; ++j) mat(i, j) = vec(i * C + j);
    } else {
        for (index_type j = 0; j < C; ++j)
            for (index_type i = 0; i < R; ++i) mat(i, j) = vec(i + R * j);
    }
}

/// Dense matrix -> flat array through an output iterator, row by row (byRows) or column by column
/// (reference include/BlitzHelpers.hpp:215-236; LAPACK call sites use the column-wise form).
template <typename T, typename OutputItr>
void reshapeMatTo1D(const matrix_type<T>& mat, OutputItr arrItr, bool byRows = true) {
    static_assert(std::is_same<typename std::iterator_traits<OutputItr>::value_type, T>::value,
                  "Matrix value type differs from array value type");
    if (byRows) {
        for (index_type i = 0; i < mat.rows(); ++i)
            for (index_type j = 0; j < mat.cols(); ++j) *arrItr++ = mat(i, j);
    } else {
        for (index_type j = 0; j < mat.cols(); ++j)
            for (index_type i = 0; i < mat.rows(); ++i) *arrItr++ = mat(i, j);
    }
}

/// Flat array -> dense matrix, the inverse of reshapeMatTo1D (reference :238-262).
template <typename T, typename InputItr>
void reshape1DToMat(InputItr arrItr, matrix_type<T>& mat, bool byRows = true) {
    static_assert(std::is_same<typename std::remove_cv<typename std::iterator_traits<InputItr>::value_type>::type, T>::value,
                  "Matrix value type differs from array value type");
    if (byRows) {
        for (index_type i = 0; i < mat.rows(); ++i)
            for (index_type j = 0; j < mat.cols(); ++j) mat(i, j) = *arrItr++;
    } else {
        for (index_type j = 0; j < mat.cols(); ++j)
            for (index_type i = 0; i < mat.rows(); ++i) mat(i, j) = *arrItr++;
    }
}

/// out(k) = vec(map(k))
template <typename T, typename U>
void applyIndexMap(const vector_type<T>& vec, const vector_type<U>& map, vector_type<T>& out) {
    for (index_type k = 0; k < map.length(0); ++k) out(k) = vec(map(k));
}

template <typename T>
T normMax(const matrix_type<T>& mat) {
    T m = 0;
    const T* p = mat.data();
    for (std::size_t i = 0; i < mat.numElements(); ++i) {
        const T a = p[i] < 0 ? -p[i] : p[i];
        if (!(a <= m)) m = a; // propagates NaN
    }
    return m;
}

} // namespace blitzdg
