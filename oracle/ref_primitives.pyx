# cython: language_level=3
"""Harness (own code) that exposes the REFERENCE's header-inline primitives.

It `cimport`s seekmer/_kmer.pxd, _coordinate.pxd, _coordinate_array.pxd and
_sequence.pxd from /root/reference where they lie (include path only; nothing
is copied) so that the reference's own compiled code can be called from
Python.  Built by oracle/build_ref.py into oracle/_ref/ (git-ignored).  Used
only to pin the oracle's primitives and to generate tests/golden/primitives.json.
The reference's .pyx modules (mapper, index, builder) are NOT built: they
import `logbook`/`tables`, which this image lacks, and no stand-ins are written.
"""
cimport libc.stdint
cimport libc.stdlib
from seekmer cimport _kmer
from seekmer cimport _coordinate
from seekmer cimport _coordinate_array
from seekmer cimport _sequence


def kmer_size():
    return _kmer.size()

def kmer_invalid():
    return _kmer.get_invalid()

def kmer_mask():
    return _kmer.mask()

def kmer_encode(bytes sequence, int offset):
    return _kmer.encode(sequence, offset)

def kmer_append(libc.stdint.uint64_t kmer, bytes base):
    cdef char b = base[0]
    return _kmer.append(kmer, b)

def kmer_prepend(libc.stdint.uint64_t kmer, bytes base):
    cdef char b = base[0]
    return _kmer.prepend(kmer, b)

def kmer_decode(libc.stdint.uint64_t kmer):
    return _kmer.decode(kmer)

def kmer_reverse_complement(libc.stdint.uint64_t kmer):
    return _kmer.reverse_complement(kmer)

def kmer_hash(libc.stdint.uint64_t kmer):
    return _kmer.hash(kmer)

def kmer_is_valid(libc.stdint.uint64_t kmer):
    return bool(_kmer.is_valid(kmer))

def two_bit_encode(bytes base):
    cdef char b = base[0]
    return _kmer._two_bit_encode(b)

def coordinate_invalid():
    cdef _coordinate.Coordinate c = _coordinate.get_invalid()
    return (c.entry, c.offset)

def coordinate_encode(int entry, int offset):
    cdef _coordinate.Coordinate c
    c.entry = entry
    c.offset = offset
    return _coordinate.encode(c)

def coordinate_decode(libc.stdint.int64_t value):
    cdef _coordinate.Coordinate c = _coordinate.decode(value)
    return (c.entry, c.offset)

def coordinate_reverse_complement(int entry, int offset):
    cdef _coordinate.Coordinate c
    c.entry = entry
    c.offset = offset
    c = _coordinate.reverse_complement(c)
    return (c.entry, c.offset)

def coordinate_is_valid(int entry, int offset):
    cdef _coordinate.Coordinate c
    c.entry = entry
    c.offset = offset
    return bool(_coordinate.is_valid(c))

def coordinate_compare(int e1, int o1, int e2, int o2):
    cdef _coordinate.Coordinate a
    cdef _coordinate.Coordinate b
    a.entry = e1
    a.offset = o1
    b.entry = e2
    b.offset = o2
    return _coordinate.compare(&a, &b)

def coordinate_array_reverse_complement(list items):
    cdef _coordinate_array.CoordinateArray arr = _coordinate_array.create(len(items))
    cdef int i
    for i in range(len(items)):
        arr.items[i].entry = items[i][0]
        arr.items[i].offset = items[i][1]
    _coordinate_array.reverse_complement(arr)
    out = [(arr.items[i].entry, arr.items[i].offset) for i in range(arr.size)]
    _coordinate_array.free(&arr)
    return out

def sequence_reverse_complement(bytes bases):
    cdef _sequence.Sequence s = _sequence.create(len(bases))
    cdef int i
    for i in range(len(bases)):
        s.bases[i] = bases[i]
    _sequence.reverse_complement(s)
    out = s.bases[:s.length]
    _sequence.free(&s)
    return out
