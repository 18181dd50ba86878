// m2s_ply.h — streaming .ply writer shared by m2s_write_ply (m2s_ply.cpp) and m2s_export_ply (m2s_records.cpp).
#pragma once
#include "../../include/m2s.h"

#include <cstdio>
#include <future>
#include <vector>

namespace m2s_ply {

constexpr size_t kChunkRows = 1u << 18;   // records per append: 24 MB in, 65 / 19 / 12 MB out (formats 0 / 1 / 2)

// Rows are encoded by a pool of threads into one of two buffers while the other one is being written to the file by a
// background task: encoding (and, for m2s_export_ply, the device-to-host copy of the next chunk) overlaps the file I/O.
class Writer {
public:
    ~Writer() { (void)close(); }
    m2s_status open(const char* path, uint64_t n_total, uint32_t format, float scale_multiplier);
    // One writer of SEVERAL that fill the same file (one per GPU rank): this one writes rows [first_row, first_row + n_rows)
    // of a file whose header announces n_total rows.  The file is created if needed and never truncated below its final
    // size; the writer of row 0 also writes the header and sets the file's length.
    m2s_status open_slice(const char* path, uint64_t n_total, uint32_t format, float scale_multiplier, uint64_t first_row, uint64_t n_rows);
    // rows that are already encoded (the device-side encoder of m2s_export_ply): `bytes` must be a whole number of rows;
    // returns once `rows` has been read completely
    m2s_status append_encoded(const uint8_t* rows, size_t n_rows);
    size_t row_bytes() const { return row_bytes_; }
    m2s_status append(const m2s_gaussian* records, size_t rows);   // returns once `records` has been read completely
    m2s_status close();                                            // flushes; M2S_ERR_IO if anything failed or rows are missing

private:
    FILE* f_ = nullptr;
    uint32_t format_ = 0;
    float sm_ = 0.0f;
    size_t row_bytes_ = 0;
    uint64_t expected_ = 0, written_ = 0;
    std::vector<uint8_t> buf_[2];
    std::future<bool> pending_;
    int cur_ = 0;
    bool ok_ = true;
};

}  // namespace m2s_ply
