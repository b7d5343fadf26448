// gob.h -- a small decoder for Go's encoding/gob wire format, enough to read what sybil
// writes with gob.NewEncoder(...).Encode(struct): table info.db (table_io.go:72-78), block
// info.db (column_store_io.go:308-358) and the int_/str_/set_ column files
// (column_store.go:22-74).  Format notes: SURVEY.md 8c "gob wire format".
//
// The decoder is schema-driven: it reads the wireType definitions that precede the value
// and builds a generic tree, so field order / omitted zero fields / extra fields in newer
// files do not matter.  Slices of Go int/uint kinds decode into flat int64 vectors (the
// record-id and value arrays are the bulk of every column file).  Interface-typed values
// do not occur in those files and are rejected.
#pragma once
#include <stdint.h>

#include <map>
#include <memory>
#include <string>
#include <vector>

namespace sybl {
const char *env(const char *name);  // (plan.h / engine.cpp: the diagnostic switches, read once)

namespace gob {

struct Value;
typedef std::shared_ptr<Value> ValuePtr;

// The int64 array of a decoded slice (record ids, values: 512 KB per column file of a full block).  As std::vector every
// decode allocated and freed it -- above malloc's mmap threshold, so every column file of every block paid an mmap, 128
// page faults and an munmap: measured on one core, a file of 65 536 one-byte varints took 0.9 ms to "decode", 0.13 ms with
// the buffers recycled.  IntBuf keeps a few released buffers per thread and does not zero what the decoder is about to
// fill.
struct IntBuf {
    int64_t *p = nullptr;   // (as int32_t / uint16_t when w is 4 / 2: data32 / data16)
    size_t n = 0, cap = 0;  // elements; capacity in int64 units
    int w = 8;              // bytes per element: 8 everywhere, except where the caller asked for narrow slices (DecodeOpts)
    IntBuf() {}
    IntBuf(const IntBuf &o) { copy_from(o); }
    IntBuf(IntBuf &&o) noexcept : p(o.p), n(o.n), cap(o.cap), w(o.w) { o.p = nullptr, o.n = o.cap = 0; }
    IntBuf &operator=(const IntBuf &o) {
        if (this != &o) copy_from(o);
        return *this;
    }
    IntBuf &operator=(IntBuf &&o) noexcept {
        if (this != &o) {
            release();
            p = o.p, n = o.n, cap = o.cap, w = o.w;
            o.p = nullptr, o.n = o.cap = 0;
        }
        return *this;
    }
    ~IntBuf() { release(); }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    // (the int64 view: w == 8)
    int64_t *data() { return p; }
    const int64_t *data() const { return p; }
    int64_t &operator[](size_t i) { return p[i]; }
    const int64_t &operator[](size_t i) const { return p[i]; }
    const int64_t *begin() const { return p; }
    const int64_t *end() const { return p + n; }
    const int32_t *data32() const { return (const int32_t *)p; }
    const uint16_t *data16() const { return (const uint16_t *)p; }
    int64_t at(size_t i) const { return w == 8 ? p[i] : w == 4 ? (int64_t)data32()[i] : (int64_t)data16()[i]; }  // any width
    // (new elements are NOT initialised; m elements of w bytes.  Inline while the buffer holds them: a bucket-encoded file
    // appends bin by bin, a thousand times per column file)
    void resize(size_t m) {
        if ((m * (size_t)w + 7) / 8 <= cap) n = m;
        else grow(m);
    }
    void assign(const int64_t *src, size_t m);
    void release();

   private:
    void copy_from(const IntBuf &o);
    void grow(size_t m);
};

// What the caller wants of the int slices (the loader: a block's record ids travel to the GPU as uint16 and its value
// deltas as int32 -- decoded straight into that form they never exist as 512 KB int64 arrays, which was most of the memory
// traffic of a load).  narrow: `Records` of a kBinVec come out as uint16 (IntBuf::w == 2) when every one of them fits, a
// top-level int slice as int32 (w == 4) when every value fits; int64 otherwise, so a reader must look at w.
// Where the top-level `Values []int64` slice of a column file lies, for a reader that wants the BYTES, not the numbers (the
// loader's GPU varint walk, gobgpu.hip): its first element, the end of the value message, the count its header announced.
struct RawInts {
    const uint8_t *p = nullptr, *end = nullptr;
    uint64_t n = 0;
    bool hit = false;
};
struct DecodeOpts {
    bool narrow = false;
    // non-null: a top-level struct's signed-int slice field named "Values" is not decoded -- *raw_values says where it lies and
    // the decode STOPS there (the fields behind it, VERSION, are not read; the tree holds the fields before it)
    RawInts *raw_values = nullptr;
    // non-null: likewise for a top-level `Bins` slice of struct{Value int; Records []uint} (SavedIntBucket, in that order): *raw_bins
    // says where the first bucket lies and how many the slice announced; the decode stops there
    RawInts *raw_bins = nullptr;
};

struct Value {
    enum Kind { kNil, kBool, kInt, kUint, kFloat, kString, kStruct, kSlice, kMap, kIntVec, kFloatVec, kBinVec };
    Kind kind = kNil;
    int64_t i = 0;    // kBool / kInt
    uint64_t u = 0;   // kUint
    double f = 0;     // kFloat
    std::string s;    // kString / bytes
    std::vector<std::pair<std::string, ValuePtr>> fields;  // kStruct (only fields present on the wire)
    std::vector<ValuePtr> items;                           // kSlice
    std::vector<std::pair<ValuePtr, ValuePtr>> entries;    // kMap
    IntBuf ints;                                           // kIntVec (kBinVec: every bin's records, back to back)
    std::vector<double> floats;                            // kFloatVec
    // kBinVec: a slice of struct{Value int; Records []int} -- the Bins of a bucket-encoded column file (SavedIntBucket /
    // SavedStrBucket / SavedSetBucket, column_store.go:46-74; up to 5000 per block) -- decoded into flat arrays instead
    // of one tree node, two shared_ptrs and three vectors per bin: bin k holds bin_val[k] and ints[bin_off[k] .. bin_off[k+1]).
    // bin_has[k]: bit 0 = Value was on the wire, bit 1 = Records was (zero-valued fields are omitted); bin_order: 1 when
    // Records precedes Value in the struct.
    std::vector<int64_t> bin_val, bin_off;
    std::vector<uint8_t> bin_has;
    int bin_order = 0;
    std::string type_name;

    const Value *field(const char *name) const;  // nullptr when the (zero-valued) field was omitted
    int64_t as_int(int64_t dflt = 0) const;
    bool as_bool() const { return as_int(0) != 0; }
};

// Decodes the first top-level value of a gob stream.  Returns false and sets err on failure.
bool decode(const uint8_t *data, size_t size, Value &out, std::string &err, const DecodeOpts *opts = nullptr);

// Reads a file, transparently gunzipping "*.gz" (or trying "<path>.gz" when <path> is missing,
// like GetFileDecoder, file_decoder.go:55-81).
bool read_file(const std::string &path, std::vector<uint8_t> &out, std::string &err);
// ... into a buffer the caller keeps between files (the loader's workers: one per thread): grown with realloc, never zeroed,
// never shrunk -- a plain file arrives with one read() straight into it (std::vector has no uninitialised resize: the vector
// form read 64 KB at a time through the stack and appended, a second copy of every byte and four system calls per file).
struct FileBuf {
    uint8_t *p = nullptr;
    size_t n = 0, cap = 0;  // bytes of the file, bytes allocated
    FileBuf() {}
    FileBuf(const FileBuf &) = delete;
    FileBuf &operator=(const FileBuf &) = delete;
    ~FileBuf();
};
bool read_file(const std::string &path, FileBuf &out, std::string &err);

// JSON rendering of a decoded tree (struct field order = wire order); used by tests.
void to_json(const Value &v, std::string &out);

}  // namespace gob
}  // namespace sybl
