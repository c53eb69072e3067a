// csrc/hdf5_io.hpp -- a dependency-free reader (and minimal writer) for the HDF5
// files the reference feeds its nets from (the HDF5 half).
//
// The reference links libhdf5 + hdf5_hl and loads every top's dataset whole with
// H5LTread_dataset_float (src/caffe/util/hdf5.cpp:10-73, called from
// HDF5DataLayer::LoadHDF5FileData, src/caffe/layers/hdf5_data_layer.cpp:27-71).
// libhdf5 is not in this image, so the subset of the HDF5 file format that h5py
// writes by default (and that the reference's own fixtures
// src/caffe/test/test_data/sample_data*.h5 use) is decoded by hand:
//   superblock v0/v1, symbol-table groups (B-tree v1 + local heap + SNOD),
//   v1 object headers with continuation blocks, simple dataspaces (v1/v2),
//   fixed-point and IEEE floating-point datatypes (little-endian, 1/2/4/8 bytes),
//   data layout v3: compact, contiguous, chunked (B-tree v1) with the
//   deflate / shuffle / fletcher32 filters (zlib for inflate).
// Anything else (v2 object headers from libver='latest', compound / string types,
// big-endian data, external storage) is reported as an error, never guessed at.
#ifndef MMS_HDF5_IO_HPP_
#define MMS_HDF5_IO_HPP_

#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace mms_h5 {

struct DatasetInfo {
  std::vector<int64_t> dims;
  int type_class = -1;   // 0 fixed-point (H5T_INTEGER), 1 floating-point (H5T_FLOAT)
  int elem_size = 0;     // bytes
  bool is_signed = false;
  int64_t count() const { int64_t c = 1; for (int64_t d : dims) c *= d; return c; }
};

class File {
 public:
  // Returns false and fills *err on any structural problem.
  bool Open(const std::string& path, std::string* err);
  std::vector<std::string> DatasetNames() const;           // every dataset, "group/member" paths for nested ones, name order
  const std::vector<std::string>& GroupNames() const { return groups_; }   // every group below the root, as paths
  bool Find(const std::string& name) const { return objects_.count(name) != 0; }
  bool Info(const std::string& name, DatasetInfo* info, std::string* err) const;
  // H5LTread_dataset_float: the whole dataset converted to float, row-major.
  bool ReadFloat(const std::string& name, DatasetInfo* info, std::vector<float>* out, std::string* err) const;

 private:
  struct Parsed;
  bool ParseObject(uint64_t addr, Parsed* p, std::string* err) const;
  bool ReadRaw(const Parsed& p, std::vector<uint8_t>* raw, std::string* err) const;
  bool WalkGroupTree(uint64_t btree, uint64_t heap_data, int depth, std::string* err, const std::string& prefix = std::string());
  bool GroupOf(uint64_t entry, uint64_t* btree, uint64_t* heap_data) const;
  bool WalkChunkTree(uint64_t node, const Parsed& p, std::vector<uint8_t>* raw, int depth, std::string* err) const;
  bool In(uint64_t off, uint64_t len) const { return off <= buf_.size() && len <= buf_.size() - off; }
  uint64_t U(uint64_t off, int bytes) const;
  std::vector<uint8_t> buf_;
  uint64_t base_ = 0;
  mutable uint64_t nodes_visited_ = 0;        // guards against cyclic B-trees in damaged files
  std::map<std::string, uint64_t> objects_;   // dataset path -> object header address
  std::vector<std::string> groups_;
};

// What h5py's `f[name] = ndarray` produces for a float32 / float64 array: a contiguous
// little-endian dataset in the root group.  Used for the driver-format fixtures
// (do_trec_qa_clean.py:237-246 writes float64 `question/answer/label/group/overlap_feat`).
struct WriteDataset {
  std::string name;
  std::vector<int64_t> dims;
  int elem_size = 4;                 // 4 = float32, 8 = float64
  std::vector<double> values;        // row-major; narrowed to float when elem_size == 4
};
bool WriteContiguous(const std::string& path, const std::vector<WriteDataset>& sets, std::string* err);

}  // namespace mms_h5
#endif  // MMS_HDF5_IO_HPP_
