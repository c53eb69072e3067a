// csrc/hdf5_io.cpp -- see hdf5_io.hpp.  Format reference: "HDF5 File Format
// Specification Version 2.0" (the published spec of libhdf5 1.8); section numbers
// below are that document's.  Pinned by tests/test_hdf5.py against the h5py-written
// fixtures the reference's own tests hold (tests/golden/ref_sample_data*.h5).
#include "hdf5_io.hpp"

#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstring>

namespace mms_h5 {
namespace {

const uint8_t kSignature[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};
const uint64_t kUndef = ~0ull;
const int kMaxDepth = 32;
const uint64_t kMaxNodes = 1u << 20;

bool fail(std::string* err, const std::string& m) {
  if (err) *err = m;
  return false;
}

struct Filter { int id; uint32_t flags; std::vector<uint32_t> client; };

}  // namespace

struct File::Parsed {
  DatasetInfo info;
  bool have_space = false, have_type = false, have_layout = false;
  int layout_class = -1;            // 0 compact, 1 contiguous, 2 chunked
  uint64_t data_addr = kUndef, data_size = 0;
  std::vector<uint8_t> compact;
  std::vector<uint64_t> chunk_dims; // rank entries (element-size entry dropped)
  uint64_t chunk_btree = kUndef;
  std::vector<Filter> filters;
};

uint64_t File::U(uint64_t off, int bytes) const {
  uint64_t v = 0;
  for (int i = 0; i < bytes; ++i) v |= (uint64_t)buf_[off + i] << (8 * i);
  return v;
}

bool File::Open(const std::string& path, std::string* err) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return fail(err, "Failed opening HDF5 file: " + path);
  std::fseek(f, 0, SEEK_END);
  const long n = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  buf_.resize(n > 0 ? (size_t)n : 0);
  const size_t got = buf_.empty() ? 0 : std::fread(buf_.data(), 1, buf_.size(), f);
  std::fclose(f);
  if (got != buf_.size()) return fail(err, "short read: " + path);
  // III.A superblock: the signature may sit at 0, 512, 1024, ... (user block)
  uint64_t sb = kUndef;
  for (uint64_t off = 0; off + 8 <= buf_.size(); off = off ? off * 2 : 512)
    if (!std::memcmp(buf_.data() + off, kSignature, 8)) { sb = off; break; }
  if (sb == kUndef) return fail(err, "not an HDF5 file (no signature): " + path);
  if (!In(sb, 96)) return fail(err, "truncated superblock");
  const int version = buf_[sb + 8];
  if (version > 1) return fail(err, "superblock version " + std::to_string(version) +
                               " (libver='latest' files) is not supported");
  if (buf_[sb + 13] != 8 || buf_[sb + 14] != 8) return fail(err, "only 8-byte offsets/lengths are supported");
  uint64_t p = sb + 24 + (version == 1 ? 4 : 0);
  base_ = U(p, 8);
  if (base_ == kUndef) base_ = 0;
  base_ += 0;  // addresses are relative to the base address (== sb for user-block files)
  p += 32;     // base, free-space, end-of-file, driver-info addresses
  // root group symbol table entry (III.C): name offset, header address, cache type, scratch
  if (!In(p, 40)) return fail(err, "truncated root symbol table entry");
  const uint64_t root_hdr = U(p + 8, 8);
  const uint32_t cache_type = (uint32_t)U(p + 16, 4);
  uint64_t btree = kUndef, heap = kUndef;
  if (cache_type == 1) {
    btree = U(p + 24, 8);
    heap = U(p + 32, 8);
  } else {
    // find the symbol-table message (0x11) in the root object header
    uint64_t a = base_ + root_hdr;
    if (!In(a, 16) || buf_[a] != 1) return fail(err, "unsupported root object header");
    const int nmsg = (int)U(a + 2, 2);
    uint64_t q = a + 16, end = q + U(a + 8, 4);
    for (int i = 0; i < nmsg && q + 8 <= end && In(q, 8); ++i) {
      const int type = (int)U(q, 2), size = (int)U(q + 2, 2);
      if (type == 0x11 && In(q + 8, 16)) { btree = U(q + 8, 8); heap = U(q + 16, 8); }
      q += 8 + size;
    }
  }
  if (btree == kUndef || heap == kUndef) return fail(err, "root group has no symbol table");
  // III.D local heap: "HEAP", version, reserved[3], data size, free-list head, data address
  const uint64_t h = base_ + heap;
  if (!In(h, 32) || std::memcmp(buf_.data() + h, "HEAP", 4)) return fail(err, "bad local heap");
  const uint64_t heap_data = base_ + U(h + 24, 8);
  return WalkGroupTree(base_ + btree, heap_data, 0, err);
}

// Is the object at `hdr` a (symbol-table) group?  Fills btree / heap-data addresses.  A link's cache type 1
// carries them in its scratch space; otherwise the object header's symbol-table message (0x11) does.
bool File::GroupOf(uint64_t entry, uint64_t* btree, uint64_t* heap_data) const {
  uint64_t bt = kUndef, hp = kUndef;
  if ((uint32_t)U(entry + 16, 4) == 1) {
    bt = U(entry + 24, 8);
    hp = U(entry + 32, 8);
  } else {
    const uint64_t a = base_ + U(entry + 8, 8);
    if (!In(a, 16) || buf_[a] != 1) return false;
    const int nmsg = (int)U(a + 2, 2);
    uint64_t q = a + 16;
    const uint64_t end = q + U(a + 8, 4);
    for (int i = 0; i < nmsg && q + 8 <= end && In(q, 8); ++i) {
      const int type = (int)U(q, 2), size = (int)U(q + 2, 2);
      if (type == 0x11 && In(q + 8, 16)) { bt = U(q + 8, 8); hp = U(q + 16, 8); }
      q += 8 + size;
    }
  }
  if (bt == kUndef || hp == kUndef) return false;
  const uint64_t h = base_ + hp;
  if (!In(h, 32) || std::memcmp(buf_.data() + h, "HEAP", 4)) return false;
  *btree = base_ + bt;
  *heap_data = base_ + U(h + 24, 8);
  return true;
}

// III.A.1 version-1 B-tree, node type 0 (group nodes); leaves point at SNODs (III.B).  Members that are
// groups themselves are walked too and their datasets recorded under "group/member" paths (the layout of the
// reference's HDF5 snapshots: /data/<layer name>/<param index>, net.cpp:797-844).
bool File::WalkGroupTree(uint64_t node, uint64_t heap_data, int depth, std::string* err, const std::string& prefix) {
  if (depth > kMaxDepth || ++nodes_visited_ > kMaxNodes) return fail(err, "group B-tree too deep or cyclic");
  if (!In(node, 24)) return fail(err, "group node outside the file");
  if (!std::memcmp(buf_.data() + node, "SNOD", 4)) {
    const int nsym = (int)U(node + 6, 2);
    if (!In(node + 8, (uint64_t)nsym * 40)) return fail(err, "truncated symbol node");
    for (int i = 0; i < nsym; ++i) {
      const uint64_t e = node + 8 + (uint64_t)i * 40;
      const uint64_t name_at = heap_data + U(e, 8);
      if (!In(name_at, 1)) return fail(err, "link name outside the heap");
      const char* s = (const char*)buf_.data() + name_at;
      const size_t maxlen = buf_.size() - name_at;
      const std::string name = prefix + std::string(s, strnlen(s, maxlen));
      uint64_t sub_btree, sub_heap;
      if (GroupOf(e, &sub_btree, &sub_heap)) {
        groups_.push_back(name);
        if (!WalkGroupTree(sub_btree, sub_heap, depth + 1, err, name + "/")) return false;
      } else {
        objects_[name] = base_ + U(e + 8, 8);
      }
    }
    return true;
  }
  if (std::memcmp(buf_.data() + node, "TREE", 4) || buf_[node + 4] != 0) return fail(err, "bad group B-tree node");
  const int nent = (int)U(node + 6, 2);
  if (!In(node + 24, (uint64_t)nent * 16 + 8)) return fail(err, "truncated group B-tree node");
  for (int i = 0; i < nent; ++i) {
    const uint64_t child = U(node + 24 + 8 + (uint64_t)i * 16, 8);   // key_i, child_i, key_{i+1}, ...
    if (!WalkGroupTree(base_ + child, heap_data, depth + 1, err, prefix)) return false;
  }
  return true;
}

std::vector<std::string> File::DatasetNames() const {
  std::vector<std::string> v;
  for (const auto& kv : objects_) v.push_back(kv.first);
  return v;
}

// IV.A.1.a version-1 object header and the messages a simple dataset carries (IV.A.2).
bool File::ParseObject(uint64_t addr, Parsed* p, std::string* err) const {
  if (!In(addr, 16)) return fail(err, "object header outside the file");
  if (!std::memcmp(buf_.data() + addr, "OHDR", 4)) return fail(err, "version-2 object headers are not supported");
  if (buf_[addr] != 1) return fail(err, "unknown object header version");
  int remaining = (int)U(addr + 2, 2);
  std::vector<std::pair<uint64_t, uint64_t>> blocks;
  blocks.push_back({addr + 16, U(addr + 8, 4)});
  for (size_t b = 0; b < blocks.size() && remaining > 0; ++b) {
    if (blocks.size() > 64) return fail(err, "too many object header continuation blocks");
    uint64_t q = blocks[b].first;
    const uint64_t end = q + blocks[b].second;
    if (!In(q, blocks[b].second)) return fail(err, "object header block outside the file");
    while (q + 8 <= end && remaining > 0) {
      const int type = (int)U(q, 2);
      const uint64_t size = U(q + 2, 2);
      const uint64_t m = q + 8;
      if (m + size > end) return fail(err, "object header message overruns its block");
      --remaining;
      q = m + size;
      switch (type) {
        case 0x0001: {  // dataspace
          const int ver = buf_[m], rank = buf_[m + 1], flags = buf_[m + 2];
          uint64_t d = m + (ver == 1 ? 8 : 4);
          if (ver != 1 && ver != 2) return fail(err, "unknown dataspace version");
          if (ver == 2 && buf_[m + 3] == 2) return fail(err, "null dataspace");
          (void)flags;
          if (d + 8ull * rank > m + size) return fail(err, "truncated dataspace message");
          p->info.dims.clear();
          for (int i = 0; i < rank; ++i) p->info.dims.push_back((int64_t)U(d + 8ull * i, 8));
          p->have_space = true;
          break;
        }
        case 0x0003: {  // datatype
          if (size < 8) return fail(err, "truncated datatype message");
          const int cls = buf_[m] & 0x0f, bits0 = buf_[m + 1];
          p->info.type_class = cls;
          p->info.elem_size = (int)U(m + 4, 4);
          if (cls == 0) {
            p->info.is_signed = (bits0 & 0x08) != 0;
          } else if (cls != 1) {
            static const char* names[] = {"H5T_INTEGER", "H5T_FLOAT", "H5T_TIME", "H5T_STRING", "H5T_BITFIELD",
                                          "H5T_OPAQUE", "H5T_COMPOUND", "H5T_REFERENCE", "H5T_ENUM", "H5T_VLEN",
                                          "H5T_ARRAY"};
            return fail(err, std::string("Unsupported datatype class: ") + (cls <= 10 ? names[cls] : "unknown"));
          }
          if (bits0 & 0x01) return fail(err, "big-endian datasets are not supported");
          const int s = p->info.elem_size;
          if (cls == 1 && s != 4 && s != 8) return fail(err, "only 4- and 8-byte IEEE floats are supported");
          if (cls == 0 && s != 1 && s != 2 && s != 4 && s != 8) return fail(err, "unsupported integer size");
          p->have_type = true;
          break;
        }
        case 0x0008: {  // data layout
          const int ver = buf_[m];
          if (ver != 3) return fail(err, "data layout message version " + std::to_string(ver) + " is not supported");
          p->layout_class = buf_[m + 1];
          if (p->layout_class == 0) {
            const uint64_t n = U(m + 2, 2);
            if (4 + n > size) return fail(err, "truncated compact layout");
            p->compact.assign(buf_.begin() + m + 4, buf_.begin() + m + 4 + n);
          } else if (p->layout_class == 1) {
            p->data_addr = U(m + 2, 8);
            p->data_size = U(m + 10, 8);
          } else if (p->layout_class == 2) {
            const int nd = buf_[m + 2];   // rank + 1 (last entry = element size)
            p->chunk_btree = U(m + 3, 8);
            if (nd < 2 || 11 + 4ull * nd > size) return fail(err, "truncated chunked layout");
            p->chunk_dims.clear();
            for (int i = 0; i < nd - 1; ++i) p->chunk_dims.push_back(U(m + 11 + 4ull * i, 4));
          } else {
            return fail(err, "unknown layout class");
          }
          p->have_layout = true;
          break;
        }
        case 0x000B: {  // filter pipeline
          const int ver = buf_[m], nf = buf_[m + 1];
          uint64_t f = m + (ver == 1 ? 8 : 2);
          if (ver != 1 && ver != 2) return fail(err, "unknown filter pipeline version");
          for (int i = 0; i < nf; ++i) {
            if (f + 8 > m + size) return fail(err, "truncated filter pipeline");
            Filter fl;
            fl.id = (int)U(f, 2);
            uint64_t name_len = 0;
            if (ver == 1 || fl.id >= 256) { name_len = U(f + 2, 2); f += 4; } else { f += 2; }
            fl.flags = (uint32_t)U(f, 2);
            const int ncd = (int)U(f + 2, 2);
            f += 4;
            f += (ver == 1) ? ((name_len + 7) & ~7ull) : name_len;
            if (f + 4ull * ncd > m + size) return fail(err, "truncated filter client data");
            for (int c = 0; c < ncd; ++c) fl.client.push_back((uint32_t)U(f + 4ull * c, 4));
            f += 4ull * ncd;
            if (ver == 1 && (ncd & 1)) f += 4;
            p->filters.push_back(fl);
          }
          break;
        }
        case 0x0010: {  // continuation
          if (size < 16) return fail(err, "truncated continuation message");
          blocks.push_back({base_ + U(m, 8), U(m + 8, 8)});
          break;
        }
        case 0x0011:
          return fail(err, "object is a group, not a dataset");
        default:
          break;  // NIL, fill value, modification time, attributes ...: not needed
      }
    }
  }
  if (!p->have_space || !p->have_type || !p->have_layout) return fail(err, "object is not a simple dataset");
  // Plausibility: a Blob counts elements in an int (blob.cpp Reshape CHECK_LE(shape[i], INT_MAX / count_)),
  // uncompressed data cannot be larger than the file and deflate expands at most ~1032x.
  uint64_t count = 1;
  for (int64_t d : p->info.dims) {
    if (d < 0 || (d > 0 && count > 0x7fffffffull / (uint64_t)d)) return fail(err, "blob size exceeds INT_MAX");
    count *= (uint64_t)d;
  }
  const uint64_t bytes = count * (uint64_t)p->info.elem_size;
  const uint64_t limit = p->layout_class == 2 && !p->filters.empty() ? buf_.size() * 1100ull : buf_.size();
  if (bytes > limit) return fail(err, "dataset larger than its file can hold");
  return true;
}

bool File::Info(const std::string& name, DatasetInfo* info, std::string* err) const {
  auto it = objects_.find(name);
  if (it == objects_.end()) return fail(err, "Failed to find HDF5 dataset " + name);
  Parsed p;
  if (!ParseObject(it->second, &p, err)) return false;
  *info = p.info;
  return true;
}

namespace {

bool apply_filters_reverse(const std::vector<Filter>& filters, uint32_t mask, int elem_size,
                           std::vector<uint8_t>* buf, size_t expect, std::string* err) {
  for (int i = (int)filters.size() - 1; i >= 0; --i) {
    if (mask & (1u << i)) continue;
    const Filter& f = filters[i];
    if (f.id == 1) {            // deflate
      std::vector<uint8_t> out(expect);
      uLongf n = (uLongf)out.size();
      const int rc = uncompress(out.data(), &n, buf->data(), (uLong)buf->size());
      if (rc != Z_OK) return fail(err, "inflate failed on a chunk (zlib rc " + std::to_string(rc) + ")");
      out.resize(n);
      buf->swap(out);
    } else if (f.id == 2) {     // shuffle: byte planes -> elements
      const size_t es = f.client.empty() ? (size_t)elem_size : f.client[0];
      const size_t ne = es ? buf->size() / es : 0;
      std::vector<uint8_t> out(*buf);
      for (size_t b = 0; b < es; ++b)
        for (size_t e = 0; e < ne; ++e) out[e * es + b] = (*buf)[b * ne + e];
      buf->swap(out);
    } else if (f.id == 3) {     // fletcher32: checksum trails the data
      if (buf->size() < 4) return fail(err, "chunk shorter than its checksum");
      buf->resize(buf->size() - 4);
    } else {
      return fail(err, "unsupported HDF5 filter id " + std::to_string(f.id));
    }
  }
  return true;
}

}  // namespace

// III.A.1 version-1 B-tree, node type 1 (raw data chunks).
bool File::WalkChunkTree(uint64_t node, const Parsed& p, std::vector<uint8_t>* raw, int depth, std::string* err) const {
  if (depth > kMaxDepth || ++nodes_visited_ > kMaxNodes) return fail(err, "chunk B-tree too deep or cyclic");
  if (!In(node, 24) || std::memcmp(buf_.data() + node, "TREE", 4) || buf_[node + 4] != 1)
    return fail(err, "bad chunk B-tree node");
  const int level = buf_[node + 5], nent = (int)U(node + 6, 2);
  const int rank = (int)p.info.dims.size();
  const uint64_t key_bytes = 8 + 8ull * (rank + 1);
  if (!In(node + 24, (uint64_t)nent * (key_bytes + 8) + key_bytes)) return fail(err, "truncated chunk B-tree node");
  const size_t es = (size_t)p.info.elem_size;
  size_t chunk_elems = 1;
  for (uint64_t c : p.chunk_dims) chunk_elems *= (size_t)c;
  for (int i = 0; i < nent; ++i) {
    const uint64_t key = node + 24 + (uint64_t)i * (key_bytes + 8);
    const uint64_t child = base_ + U(key + key_bytes, 8);
    if (level > 0) {
      if (!WalkChunkTree(child, p, raw, depth + 1, err)) return false;
      continue;
    }
    const uint64_t nbytes = U(key, 4);
    const uint32_t mask = (uint32_t)U(key + 4, 4);
    std::vector<uint64_t> off(rank);
    for (int d = 0; d < rank; ++d) off[d] = U(key + 8 + 8ull * d, 8);
    if (!In(child, nbytes)) return fail(err, "chunk outside the file");
    std::vector<uint8_t> chunk(buf_.begin() + child, buf_.begin() + child + nbytes);
    if (!apply_filters_reverse(p.filters, mask, p.info.elem_size, &chunk, chunk_elems * es, err)) return false;
    if (chunk.size() != chunk_elems * es) return fail(err, "chunk has an unexpected size");
    // scatter the chunk's rows (last axis contiguous) into the dataset, clipping edge chunks
    if (rank == 0) return fail(err, "chunked scalar dataset");
    const uint64_t last = (uint64_t)p.info.dims[rank - 1];
    if (off[rank - 1] >= last) continue;
    const uint64_t run = std::min<uint64_t>(p.chunk_dims[rank - 1], last - off[rank - 1]);
    const size_t rows = chunk_elems / (size_t)p.chunk_dims[rank - 1];
    std::vector<uint64_t> idx(rank, 0);
    for (size_t r = 0; r < rows; ++r) {
      size_t rem = r;
      bool inside = true;
      uint64_t lin = 0;
      for (int d = rank - 2; d >= 0; --d) { idx[d] = rem % p.chunk_dims[d]; rem /= p.chunk_dims[d]; }
      for (int d = 0; d < rank - 1; ++d) {
        const uint64_t g = off[d] + idx[d];
        if (g >= (uint64_t)p.info.dims[d]) { inside = false; break; }
        lin = lin * (uint64_t)p.info.dims[d] + g;
      }
      if (!inside) continue;
      lin = lin * last + off[rank - 1];
      std::memcpy(raw->data() + lin * es, chunk.data() + r * (size_t)p.chunk_dims[rank - 1] * es, run * es);
    }
  }
  return true;
}

bool File::ReadRaw(const Parsed& p, std::vector<uint8_t>* raw, std::string* err) const {
  const uint64_t bytes = (uint64_t)p.info.count() * (uint64_t)p.info.elem_size;
  raw->assign(bytes, 0);
  if (p.layout_class == 0) {
    if (p.compact.size() < bytes) return fail(err, "compact dataset shorter than its dataspace");
    std::memcpy(raw->data(), p.compact.data(), bytes);
  } else if (p.layout_class == 1) {
    if (p.data_addr == kUndef) return true;  // never written: fill value (0)
    if (!In(base_ + p.data_addr, bytes)) return fail(err, "contiguous dataset outside the file");
    std::memcpy(raw->data(), buf_.data() + base_ + p.data_addr, bytes);
  } else {
    if (p.chunk_dims.size() != p.info.dims.size()) return fail(err, "chunk rank differs from dataset rank");
    uint64_t chunk_bytes = (uint64_t)p.info.elem_size;
    for (uint64_t c : p.chunk_dims) {
      if (c == 0) return fail(err, "zero chunk dimension");
      if (chunk_bytes > 0x7fffffffull / c) return fail(err, "implausible chunk size");
      chunk_bytes *= c;
    }
    if (chunk_bytes > buf_.size() * 1100ull) return fail(err, "chunk larger than its file can hold");
    nodes_visited_ = 0;
    if (p.chunk_btree == kUndef) return true;
    return WalkChunkTree(base_ + p.chunk_btree, p, raw, 0, err);
  }
  return true;
}

bool File::ReadFloat(const std::string& name, DatasetInfo* info, std::vector<float>* out, std::string* err) const {
  auto it = objects_.find(name);
  if (it == objects_.end()) return fail(err, "Failed to find HDF5 dataset " + name);
  Parsed p;
  if (!ParseObject(it->second, &p, err)) return false;
  std::vector<uint8_t> raw;
  if (!ReadRaw(p, &raw, err)) return false;
  const size_t n = (size_t)p.info.count();
  out->resize(n);
  const uint8_t* r = raw.data();
#define MMS_CONV(T) for (size_t i = 0; i < n; ++i) { T v; std::memcpy(&v, r + i * sizeof(T), sizeof(T)); (*out)[i] = (float)v; }
  if (p.info.type_class == 1) {
    if (p.info.elem_size == 4) std::memcpy(out->data(), r, n * 4);
    else MMS_CONV(double)
  } else if (p.info.is_signed) {
    switch (p.info.elem_size) {
      case 1: MMS_CONV(int8_t) break;
      case 2: MMS_CONV(int16_t) break;
      case 4: MMS_CONV(int32_t) break;
      default: MMS_CONV(int64_t) break;
    }
  } else {
    switch (p.info.elem_size) {
      case 1: MMS_CONV(uint8_t) break;
      case 2: MMS_CONV(uint16_t) break;
      case 4: MMS_CONV(uint32_t) break;
      default: MMS_CONV(uint64_t) break;
    }
  }
#undef MMS_CONV
  if (info) *info = p.info;
  return true;
}

// ------------------------------------------------------------------ writer
namespace {
struct Out {
  std::vector<uint8_t> b;
  void u(uint64_t v, int bytes) { for (int i = 0; i < bytes; ++i) b.push_back((uint8_t)(v >> (8 * i))); }
  void pad8() { while (b.size() & 7) b.push_back(0); }
  void patch(size_t at, uint64_t v, int bytes) { for (int i = 0; i < bytes; ++i) b[at + i] = (uint8_t)(v >> (8 * i)); }
};
}  // namespace

namespace {
struct GroupNode {
  std::map<std::string, const WriteDataset*> sets;    // leaf name -> dataset
  std::map<std::string, GroupNode> groups;            // member groups
};
struct Emitted { uint64_t hdr, btree, heap; bool group; };

Emitted emit_dataset(Out& o, const WriteDataset& d) {
  o.pad8();
  Emitted e{o.b.size(), 0, 0, false};
  const int rank = (int)d.dims.size();
  const uint64_t space = 8 + 8ull * rank, type = 8 + 12, layout = 24;
  o.u(1, 1); o.u(0, 1); o.u(3, 2); o.u(1, 4); o.u((8 + space) + (8 + type + 4) + (8 + layout), 4); o.u(0, 4);
  o.u(1, 2); o.u(space, 2); o.u(0, 1); o.u(0, 3);                 // dataspace v1, no max dims
  o.u(1, 1); o.u(rank, 1); o.u(0, 1); o.u(0, 5);
  for (int64_t x : d.dims) o.u((uint64_t)x, 8);
  o.u(3, 2); o.u(type + 4, 2); o.u(1, 1); o.u(0, 3);              // datatype: IEEE float, LE (constant message)
  o.u(0x11, 1); o.u(0x20, 1); o.u(d.elem_size == 4 ? 31 : 63, 1); o.u(0, 1); o.u(d.elem_size, 4);
  if (d.elem_size == 4) { o.u(0, 2); o.u(32, 2); o.u(23, 1); o.u(8, 1); o.u(0, 1); o.u(23, 1); o.u(127, 4); }
  else                  { o.u(0, 2); o.u(64, 2); o.u(52, 1); o.u(11, 1); o.u(0, 1); o.u(52, 1); o.u(1023, 4); }
  o.u(0, 4);
  o.u(8, 2); o.u(layout, 2); o.u(0, 1); o.u(0, 3);                // layout v3 contiguous
  o.u(3, 1); o.u(1, 1);
  const size_t addr_at = o.b.size(); o.u(0, 8);
  const uint64_t bytes = (uint64_t)d.values.size() * d.elem_size;
  o.u(bytes, 8); o.u(0, 6);
  o.pad8();
  o.patch(addr_at, o.b.size(), 8);
  for (double v : d.values) {
    if (d.elem_size == 4) { const float f = (float)v; uint32_t w; std::memcpy(&w, &f, 4); o.u(w, 4); }
    else { uint64_t w; std::memcpy(&w, &v, 8); o.u(w, 8); }
  }
  return e;
}

// members first (their addresses are needed by the symbol node), then this group's object header (one
// symbol-table message), local heap, one B-tree node and its symbol nodes (8 members each, up to 256 in all)
bool emit_group(Out& o, const GroupNode& g, Emitted* out, std::string* err) {
  struct Member { std::string name; Emitted e; };
  std::vector<Member> mem;
  for (const auto& kv : g.sets) mem.push_back({kv.first, emit_dataset(o, *kv.second)});
  for (const auto& kv : g.groups) {
    Emitted e;
    if (!emit_group(o, kv.second, &e, err)) return false;
    mem.push_back({kv.first, e});
  }
  std::sort(mem.begin(), mem.end(), [](const Member& a, const Member& b) { return a.name < b.name; });
  // one B-tree node (internal K = 16: up to 32 children) over symbol nodes of up to 8 entries (leaf K = 4)
  if (mem.empty() || mem.size() > 256) return fail(err, "the HDF5 writer handles 1..256 members per group (one B-tree node over 32 symbol nodes)");
  const size_t nsnod = (mem.size() + 7) / 8;
  o.pad8();
  Emitted me{o.b.size(), 0, 0, true};
  o.u(1, 1); o.u(0, 1); o.u(1, 2); o.u(1, 4); o.u(24, 4); o.u(0, 4);
  o.u(0x11, 2); o.u(16, 2); o.u(0, 1); o.u(0, 3);
  const size_t stm_at = o.b.size(); o.u(0, 8); o.u(0, 8);
  std::vector<uint64_t> name_off;
  std::vector<uint8_t> heap(8, 0);                               // offset 0 is the empty string
  for (const Member& m : mem) {
    name_off.push_back(heap.size());
    heap.insert(heap.end(), m.name.begin(), m.name.end());
    heap.push_back(0);
    while (heap.size() & 7) heap.push_back(0);
  }
  me.heap = o.b.size();
  o.b.insert(o.b.end(), {'H', 'E', 'A', 'P'});
  o.u(0, 1); o.u(0, 3); o.u(heap.size(), 8); o.u(kUndef, 8); o.u(me.heap + 32, 8);
  o.b.insert(o.b.end(), heap.begin(), heap.end());
  me.btree = o.b.size();
  o.b.insert(o.b.end(), {'T', 'R', 'E', 'E'});
  o.u(0, 1); o.u(0, 1); o.u(nsnod, 2); o.u(kUndef, 8); o.u(kUndef, 8);
  o.u(0, 8);                                                     // key 0: the empty string
  std::vector<size_t> child_at;
  for (size_t c = 0; c < nsnod; ++c) {                           // child c, then key c+1 = its LAST (largest) name
    child_at.push_back(o.b.size()); o.u(0, 8);
    const size_t last = std::min(mem.size(), 8 * (c + 1)) - 1;
    o.u(name_off[last], 8);
  }
  for (size_t c = nsnod; c < 2 * 16; ++c) { o.u(0, 8); o.u(0, 8); }   // unused child/key slots (2K = 32 entries)
  o.patch(stm_at, me.btree, 8); o.patch(stm_at + 8, me.heap, 8);
  for (size_t c = 0; c < nsnod; ++c) {
    o.patch(child_at[c], o.b.size(), 8);
    const size_t first = 8 * c, cnt = std::min<size_t>(8, mem.size() - first);
    o.b.insert(o.b.end(), {'S', 'N', 'O', 'D'});
    o.u(1, 1); o.u(0, 1); o.u(cnt, 2);
    for (size_t i = 0; i < 8; ++i) {
      if (i < cnt) {
        const Member& m = mem[first + i];
        o.u(name_off[first + i], 8); o.u(m.e.hdr, 8);
        if (m.e.group) { o.u(1, 4); o.u(0, 4); o.u(m.e.btree, 8); o.u(m.e.heap, 8); }
        else { o.u(0, 4); o.u(0, 4); o.u(0, 8); o.u(0, 8); }
      } else {
        o.u(0, 8); o.u(0, 8); o.u(0, 4); o.u(0, 4); o.u(0, 8); o.u(0, 8);
      }
    }
  }
  *out = me;
  return true;
}
}  // namespace

// Dataset names may be paths ("data/sim/0"): the groups along the way are created (symbol-table groups, v1
// object headers -- what libhdf5 1.8 writes by default and what Net::ToHDF5 produces, net.cpp:893-960).
bool WriteContiguous(const std::string& path, const std::vector<WriteDataset>& sets, std::string* err) {
  GroupNode root;
  for (const WriteDataset& d : sets) {
    int64_t c = 1;
    for (int64_t x : d.dims) c *= x;
    if ((int64_t)d.values.size() != c || (d.elem_size != 4 && d.elem_size != 8) || d.name.empty() ||
        d.name.front() == '/' || d.name.back() == '/')
      return fail(err, "bad dataset description: " + d.name);
    GroupNode* g = &root;
    size_t at = 0, slash;
    while ((slash = d.name.find('/', at)) != std::string::npos) {
      if (slash == at) return fail(err, "bad dataset path: " + d.name);
      g = &g->groups[d.name.substr(at, slash - at)];
      at = slash + 1;
    }
    if (!g->sets.emplace(d.name.substr(at), &d).second) return fail(err, "duplicate dataset: " + d.name);
  }
  if (sets.empty()) return fail(err, "nothing to write");
  Out o;
  // superblock v0 (96 bytes incl. the root symbol table entry); addresses patched below
  o.b.insert(o.b.end(), kSignature, kSignature + 8);
  o.u(0, 1); o.u(0, 1); o.u(0, 1); o.u(0, 1); o.u(0, 1); o.u(8, 1); o.u(8, 1); o.u(0, 1);
  o.u(4, 2); o.u(16, 2); o.u(0, 4);
  o.u(0, 8); o.u(kUndef, 8);
  const size_t eof_at = o.b.size(); o.u(0, 8);
  o.u(kUndef, 8);
  o.u(0, 8);                                   // root: link name offset
  const size_t root_hdr_at = o.b.size(); o.u(0, 8);
  o.u(1, 4); o.u(0, 4);                        // cache type 1: scratch = btree, heap
  const size_t scr_at = o.b.size(); o.u(0, 8); o.u(0, 8);
  Emitted r;
  if (!emit_group(o, root, &r, err)) return false;
  o.patch(root_hdr_at, r.hdr, 8);
  o.patch(scr_at, r.btree, 8); o.patch(scr_at + 8, r.heap, 8);
  o.patch(eof_at, o.b.size(), 8);
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) return fail(err, "cannot write " + path);
  const size_t n = std::fwrite(o.b.data(), 1, o.b.size(), f);
  std::fclose(f);
  return n == o.b.size() ? true : fail(err, "short write " + path);
}

}  // namespace mms_h5
