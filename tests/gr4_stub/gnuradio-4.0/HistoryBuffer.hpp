// TEST-ONLY stand-in (see Block.hpp) for gr::HistoryBuffer as the reference's FIR blocks use it
// (symbol_filter.hpp:43,95-103,211-214; interpolating_fir_filter.hpp:34,65-72,94-99; pfb_arb_resampler.hpp:105-113,
// 136-152): capacity a power of two, push_back() puts the newest item at index 0, cbegin() is a CONTIGUOUS
// newest -> oldest range (std::inner_product walks it), push_back_bulk(range), size(), operator[] (mutable for
// syncword_detection.hpp:294), copyable.
#pragma once
#include <cstddef>
#include <vector>

namespace gr {
template <typename T>
class HistoryBuffer
{
    size_t _cap = 1, _size = 0, _w = 0; // storage is mirrored: item i (0 = newest) lives at _w + i and _w + i + _cap
    std::vector<T> _buf;

public:
    explicit HistoryBuffer(size_t capacity = 1) : _cap(capacity ? capacity : 1), _w(_cap), _buf(2 * _cap) {}
    void push_back(const T& v)
    {
        _w = _w == 0 ? _cap - 1 : _w - 1;
        _buf[_w] = v;
        _buf[_w + _cap] = v;
        if (_size < _cap) ++_size;
    }
    template <typename R>
    void push_back_bulk(const R& range)
    {
        for (const auto& v : range) push_back(v);
    }
    size_t size() const { return _size; }
    size_t capacity() const { return _cap; }
    const T* cbegin() const { return _buf.data() + (_w % _cap); }
    const T* cend() const { return cbegin() + _size; }
    const T* begin() const { return cbegin(); }
    const T* end() const { return cend(); }
    // indexed access goes to ONE of the two copies, the same one for reading and writing: syncword_detection.hpp:294
    // sets `_history[i].detection = true` and reads the flag back hundreds of items later (a mirrored copy that is
    // read on one side and written on the other loses detections)
    const T& operator[](size_t i) const { return _buf[(_w + i) % _cap]; }
    T& operator[](size_t i) { return _buf[(_w + i) % _cap]; }
};
} // namespace gr
