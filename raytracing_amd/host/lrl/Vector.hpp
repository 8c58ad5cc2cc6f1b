// Growable array with the member names the reference apps use on the author's (absent) Vector<T>:
// `.data`, `.length`, `pushBack`, `operator+=` with an initializer list or another Vector,
// `operator[]` (use sites: BVH/BVH.cpp:189,280-379,391; Pool/Pool.cpp:242-298,515-605).
// Growth policy = capacity doubling from 1 (BVH.exe@0x140007988).
#pragma once
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <new>
#include <utility>

namespace lrl {

template <class T> struct Vector {
    T* data = nullptr;
    unsigned int length = 0;
    unsigned int lengthAll = 0;

    Vector() = default;
    Vector(std::initializer_list<T> l) { for (auto const& v : l) pushBack(v); }
    Vector(Vector const& o) { *this = o; }
    Vector(Vector&& o) noexcept : data(o.data), length(o.length), lengthAll(o.lengthAll) { o.data = nullptr; o.length = o.lengthAll = 0; }
    ~Vector() { release(); }

    Vector& operator=(Vector const& o) {
        if (this == &o) return *this;
        clear();
        reserve(o.length);
        for (unsigned int i = 0; i < o.length; ++i) new (data + i) T(o.data[i]);
        length = o.length;
        return *this;
    }
    Vector& operator=(Vector&& o) noexcept {
        if (this == &o) return *this;
        release();
        data = o.data; length = o.length; lengthAll = o.lengthAll;
        o.data = nullptr; o.length = o.lengthAll = 0;
        return *this;
    }
    void reserve(unsigned int n) {
        if (n <= lengthAll) return;
        unsigned int cap = lengthAll ? lengthAll : 1;
        while (cap < n) cap *= 2;
        T* nd = static_cast<T*>(::operator new(sizeof(T) * size_t(cap), std::align_val_t(alignof(T) < 16 ? 16 : alignof(T))));
        for (unsigned int i = 0; i < length; ++i) { new (nd + i) T(std::move(data[i])); data[i].~T(); }
        if (data) ::operator delete(data, std::align_val_t(alignof(T) < 16 ? 16 : alignof(T)));
        data = nd; lengthAll = cap;
    }
    Vector& pushBack(T const& v) {
        if (length + 1 > lengthAll) { T tmp(v); reserve(length + 1); new (data + length) T(std::move(tmp)); }  // v may alias data
        else new (data + length) T(v);
        ++length;
        return *this;
    }
    Vector& operator+=(std::initializer_list<T> l) { for (auto const& v : l) pushBack(v); return *this; }
    Vector& operator+=(Vector const& o) { reserve(length + o.length); for (unsigned int i = 0; i < o.length; ++i) pushBack(o.data[i]); return *this; }
    T& operator[](unsigned int i) { return data[i]; }
    T const& operator[](unsigned int i) const { return data[i]; }
    void clear() { for (unsigned int i = 0; i < length; ++i) data[i].~T(); length = 0; }
    T* begin() { return data; }
    T const* begin() const { return data; }
    T* end() { return data + length; }
    T const* end() const { return data + length; }

   private:
    void release() {
        clear();
        if (data) ::operator delete(data, std::align_val_t(alignof(T) < 16 ? 16 : alignof(T)));
        data = nullptr; lengthAll = 0;
    }
};

}  // namespace lrl
