// Minimal read-only safetensors reader (mmap + header parse), no third-party code.
//
// Replaces, for the T3 path, the reference's R/safetensors.R:7-10 wrapper around the
// CRAN `safetensors` package: the unchanged HF file (t3_cfg.safetensors) is read by
// the library itself so no R-torch tensor ever crosses the C ABI.
//
// File format: u64 little-endian header length N, N bytes of JSON
//   { "<name>": {"dtype":"F32","shape":[..],"data_offsets":[begin,end]}, ..., "__metadata__": {...} }
// followed by the raw little-endian tensor bytes.
#pragma once
#include <cstddef>
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace t3mi {

enum class StDtype { F32, F16, BF16, I64, I32, OTHER };

struct StTensor {
    StDtype dtype = StDtype::OTHER;
    std::vector<int64_t> shape;
    const uint8_t* data = nullptr;   // points into the mmap
    size_t nbytes = 0;
    int64_t numel() const {   // shapes are validated at open(): no negative extent, product below 2^62
        int64_t n = 1;
        for (auto s : shape) n *= s;
        return n;
    }
};

class SafeTensorsFile {
public:
    SafeTensorsFile() = default;
    ~SafeTensorsFile();
    SafeTensorsFile(const SafeTensorsFile&) = delete;
    SafeTensorsFile& operator=(const SafeTensorsFile&) = delete;

    // returns false and fills err on failure
    bool open(const std::string& path, std::string& err);
    const StTensor* find(const std::string& name) const;
    const std::map<std::string, StTensor>& tensors() const { return tensors_; }
    size_t file_size() const { return map_len_; }
    const uint8_t* base() const { return static_cast<const uint8_t*>(map_); }   // the mapped file (nullptr before open)
    int64_t mtime_ns() const { return mtime_ns_; }

    // Convert tensor `t` (F32/F16/BF16) to float32 into dst[numel].
    static bool to_f32(const StTensor& t, float* dst, std::string& err);

private:
    void* map_ = nullptr;
    size_t map_len_ = 0;
    int fd_ = -1;
    int64_t mtime_ns_ = 0;
    std::map<std::string, StTensor> tensors_;
};

}  // namespace t3mi
