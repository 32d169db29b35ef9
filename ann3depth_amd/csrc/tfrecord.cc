// tfrecord.cc — host side of the dataset plugin: CRC32C, TFRecord framing, and a hand-written protobuf wire
// parser/writer for the 8-feature tf.train.Example that ann3depth stores (reference: src/data.py:62-86 reads it,
// tools/data_tf_converter.py:27-53 writes it).  No TensorFlow, no libprotobuf.
#include <algorithm>
#include <cstring>
#include <initializer_list>

#include "a3d_internal.h"

namespace {

uint32_t g_table[8][256];
bool g_table_ready = false;

void init_table() {
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1) ? 0x82F63B78u : 0u);
    g_table[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; ++i)
    for (int t = 1; t < 8; ++t) g_table[t][i] = (g_table[t - 1][i] >> 8) ^ g_table[0][g_table[t - 1][i] & 0xFF];
  g_table_ready = true;
}

struct TableInit {
  TableInit() { init_table(); }
} g_table_init;

// slicing-by-8
uint32_t crc32c_sw(const uint8_t* p, size_t n, uint32_t crc) {
  if (!g_table_ready) init_table();
  crc = ~crc;
  while (n && (reinterpret_cast<uintptr_t>(p) & 7)) {
    crc = g_table[0][(crc ^ *p++) & 0xFF] ^ (crc >> 8);
    --n;
  }
  while (n >= 8) {
    uint64_t v;
    memcpy(&v, p, 8);
    v ^= crc;
    crc = g_table[7][v & 0xFF] ^ g_table[6][(v >> 8) & 0xFF] ^ g_table[5][(v >> 16) & 0xFF] ^
          g_table[4][(v >> 24) & 0xFF] ^ g_table[3][(v >> 32) & 0xFF] ^ g_table[2][(v >> 40) & 0xFF] ^
          g_table[1][(v >> 48) & 0xFF] ^ g_table[0][(v >> 56) & 0xFF];
    p += 8;
    n -= 8;
  }
  while (n--) crc = g_table[0][(crc ^ *p++) & 0xFF] ^ (crc >> 8);
  return ~crc;
}

#if defined(__x86_64__)
// crc32q has a latency of three cycles and a throughput of one: a single chain checks 8 bytes per 3 cycles (~8 GB/s), and
// a 4.9 MB record costs a reader thread 0.6 ms of CRC alone.  Three independent chains over three consecutive 8 KiB
// pieces fill the pipeline; the pieces' registers are joined with the "append N zero bytes" operator of the CRC (a GF(2)
// matrix, applied through four 256-entry tables) — the construction published with zlib's crc32_combine, for the
// Castagnoli polynomial.
constexpr size_t kCrcPiece = 8192;          // a power of two (the operator is built by repeated squaring)
uint32_t g_shift[4][256];                   // register after kCrcPiece zero bytes, per byte of the register
bool g_shift_ready = false;

uint32_t gf2_times(const uint32_t* mat, uint32_t vec) {
  uint32_t sum = 0;
  for (; vec; vec >>= 1, ++mat)
    if (vec & 1) sum ^= *mat;
  return sum;
}
void gf2_square(uint32_t* square, const uint32_t* mat) {
  for (int n = 0; n < 32; ++n) square[n] = gf2_times(mat, mat[n]);
}
void init_shift() {
  uint32_t even[32], odd[32];
  odd[0] = 0x82F63B78u;                     // operator for one zero BIT: the reflected polynomial, then the shifts
  for (int n = 1; n < 32; ++n) odd[n] = 1u << (n - 1);
  gf2_square(even, odd);                    // two zero bits
  gf2_square(odd, even);                    // four
  size_t len = kCrcPiece;                   // each further squaring doubles: 1 byte, 2 bytes, ...
  uint32_t* cur = odd;
  uint32_t* nxt = even;
  for (;;) {
    gf2_square(nxt, cur);                   // first pass: one zero byte
    len >>= 1;
    uint32_t* t = cur; cur = nxt; nxt = t;
    if (len == 0) break;
  }
  for (uint32_t n = 0; n < 256; ++n)
    for (int b = 0; b < 4; ++b) g_shift[b][n] = gf2_times(cur, n << (8 * b));
  g_shift_ready = true;
}
struct ShiftInit {
  ShiftInit() { init_shift(); }
} g_shift_init;
inline uint32_t shift_piece(uint32_t c) {
  return g_shift[0][c & 0xFF] ^ g_shift[1][(c >> 8) & 0xFF] ^ g_shift[2][(c >> 16) & 0xFF] ^ g_shift[3][c >> 24];
}

__attribute__((target("sse4.2"))) uint32_t crc32c_hw(const uint8_t* p, size_t n, uint32_t crc) {
  if (!g_shift_ready) init_shift();
  uint64_t c = ~crc;
  while (n && (reinterpret_cast<uintptr_t>(p) & 7)) {
    c = __builtin_ia32_crc32qi((uint32_t)c, *p++);
    --n;
  }
  while (n >= 3 * kCrcPiece) {              // three chains, joined: c0 || piece1 || piece2
    uint64_t c1 = 0, c2 = 0;
    const uint8_t* end = p + kCrcPiece;
    do {
      uint64_t v0, v1, v2;
      memcpy(&v0, p, 8);
      memcpy(&v1, p + kCrcPiece, 8);
      memcpy(&v2, p + 2 * kCrcPiece, 8);
      c = __builtin_ia32_crc32di(c, v0);
      c1 = __builtin_ia32_crc32di(c1, v1);
      c2 = __builtin_ia32_crc32di(c2, v2);
      p += 8;
    } while (p < end);
    c = shift_piece((uint32_t)c) ^ c1;
    c = shift_piece((uint32_t)c) ^ c2;
    p += 2 * kCrcPiece;
    n -= 3 * kCrcPiece;
  }
  while (n >= 8) {
    uint64_t v;
    memcpy(&v, p, 8);
    c = __builtin_ia32_crc32di(c, v);
    p += 8;
    n -= 8;
  }
  while (n--) c = __builtin_ia32_crc32qi((uint32_t)c, *p++);
  return ~(uint32_t)c;
}
bool have_sse42() { return __builtin_cpu_supports("sse4.2"); }
#else
uint32_t crc32c_hw(const uint8_t* p, size_t n, uint32_t crc) { return crc32c_sw(p, n, crc); }
bool have_sse42() { return false; }
#endif

inline uint32_t mask_crc(uint32_t c) { return ((c >> 15) | (c << 17)) + 0xA282EAD8u; }

// ---- protobuf wire helpers ----
struct Cursor {
  const uint8_t* p;
  const uint8_t* end;
};

bool read_varint(Cursor& c, uint64_t* v) {
  uint64_t r = 0;
  for (int shift = 0; shift < 64 && c.p < c.end; shift += 7) {
    uint8_t b = *c.p++;
    r |= (uint64_t)(b & 0x7F) << shift;
    if (!(b & 0x80)) {
      *v = r;
      return true;
    }
  }
  return false;
}

// reads one field; for length-delimited returns the sub-range in *sub, for varint the value in *val
bool read_field(Cursor& c, uint32_t* field, uint32_t* wt, uint64_t* val, Cursor* sub) {
  uint64_t key;
  if (!read_varint(c, &key)) return false;
  *field = (uint32_t)(key >> 3);
  *wt = (uint32_t)(key & 7);
  switch (*wt) {
    case 0: return read_varint(c, val);
    case 1: if (c.end - c.p < 8) return false; c.p += 8; return true;
    case 5: if (c.end - c.p < 4) return false; c.p += 4; return true;
    case 2: {
      uint64_t n;
      if (!read_varint(c, &n) || (uint64_t)(c.end - c.p) < n) return false;
      sub->p = c.p;
      sub->end = c.p + n;
      c.p += n;
      return true;
    }
    default: return false;
  }
}

size_t varint_size(uint64_t v) {
  size_t n = 1;
  while (v >= 0x80) { v >>= 7; ++n; }
  return n;
}
uint8_t* put_varint(uint8_t* p, uint64_t v) {
  while (v >= 0x80) { *p++ = (uint8_t)(v | 0x80); v >>= 7; }
  *p++ = (uint8_t)v;
  return p;
}
size_t ld_size(size_t payload) { return 1 + varint_size(payload) + payload; }   // field numbers < 16: 1-byte key
uint8_t* put_ld_head(uint8_t* p, uint32_t field, size_t payload) {
  *p++ = (uint8_t)((field << 3) | 2);
  return put_varint(p, payload);
}

}  // namespace

extern "C" {

uint32_t a3d_crc32c(const void* data, size_t len) {
  static const bool hw = have_sse42();
  const uint8_t* p = static_cast<const uint8_t*>(data);
  return hw ? crc32c_hw(p, len, 0) : crc32c_sw(p, len, 0);
}

uint32_t a3d_masked_crc32c(const void* data, size_t len) { return mask_crc(a3d_crc32c(data, len)); }

int a3d_tfrecord_next(const uint8_t* buf, size_t len, int verify_crc, size_t* payload_off, size_t* payload_len,
                      size_t* consumed) {
  if (!buf || !payload_off || !payload_len || !consumed) return a3d::set_error(A3D_EINVAL, "tfrecord_next: null argument");
  if (len < 12) return a3d::set_error(A3D_EFORMAT, "tfrecord: truncated header (%zu bytes)", len);
  uint64_t n;
  uint32_t hcrc;
  memcpy(&n, buf, 8);
  memcpy(&hcrc, buf + 8, 4);
  if (hcrc != a3d_masked_crc32c(buf, 8)) return a3d::set_error(A3D_EFORMAT, "tfrecord: corrupt length field");
  if (n > len - 12 || len - 12 - n < 4) return a3d::set_error(A3D_EFORMAT, "tfrecord: truncated record (%llu payload bytes)", (unsigned long long)n);
  if (verify_crc) {
    uint32_t pcrc;
    memcpy(&pcrc, buf + 12 + n, 4);
    if (pcrc != a3d_masked_crc32c(buf + 12, n)) return a3d::set_error(A3D_EFORMAT, "tfrecord: corrupt payload");
  }
  *payload_off = 12;
  *payload_len = (size_t)n;
  *consumed = 16 + (size_t)n;
  return A3D_OK;
}

int a3d_example_parse(const uint8_t* payload, size_t len, a3d_example_view* out) {
  if (!payload || !out) return a3d::set_error(A3D_EINVAL, "example_parse: null argument");
  memset(out, 0, sizeof(*out));
  int64_t* const ints[6] = {&out->image_height, &out->image_width, &out->image_channels,
                            &out->depth_height, &out->depth_width, &out->depth_channels};
  static const char* const int_names[6] = {"image_height", "image_width", "image_channels",
                                           "depth_height", "depth_width", "depth_channels"};
  unsigned seen = 0;
  Cursor ex{payload, payload + len};
  uint32_t f, wt;
  uint64_t val;
  Cursor features, entry, sub;
  while (ex.p < ex.end) {
    if (!read_field(ex, &f, &wt, &val, &features)) return a3d::set_error(A3D_EFORMAT, "example: bad wire data");
    if (f != 1 || wt != 2) continue;                       // Example.features
    while (features.p < features.end) {
      if (!read_field(features, &f, &wt, &val, &entry)) return a3d::set_error(A3D_EFORMAT, "features: bad wire data");
      if (f != 1 || wt != 2) continue;                     // Features.feature map entry
      Cursor key{nullptr, nullptr}, feat{nullptr, nullptr};
      while (entry.p < entry.end) {
        if (!read_field(entry, &f, &wt, &val, &sub)) return a3d::set_error(A3D_EFORMAT, "map entry: bad wire data");
        if (wt != 2) continue;
        if (f == 1) key = sub; else if (f == 2) feat = sub;
      }
      if (!key.p || !feat.p) continue;
      const size_t klen = key.end - key.p;
      while (feat.p < feat.end) {                          // Feature oneof
        Cursor list;
        uint32_t kind;
        if (!read_field(feat, &kind, &wt, &val, &list)) return a3d::set_error(A3D_EFORMAT, "feature: bad wire data");
        if (wt != 2) continue;
        if (kind == 1) {                                   // BytesList
          while (list.p < list.end) {
            if (!read_field(list, &f, &wt, &val, &sub)) return a3d::set_error(A3D_EFORMAT, "bytes_list: bad wire data");
            if (f != 1 || wt != 2) continue;
            if (klen == 5 && !memcmp(key.p, "image", 5)) { out->image = sub.p; out->image_bytes = sub.end - sub.p; seen |= 64; }
            else if (klen == 5 && !memcmp(key.p, "depth", 5)) { out->depth = sub.p; out->depth_bytes = sub.end - sub.p; seen |= 128; }
            break;
          }
        } else if (kind == 3) {                            // Int64List: packed (wt 2) or repeated varint (wt 0)
          while (list.p < list.end) {
            if (!read_field(list, &f, &wt, &val, &sub)) return a3d::set_error(A3D_EFORMAT, "int64_list: bad wire data");
            if (f != 1) continue;
            if (wt == 2 && !read_varint(sub, &val)) return a3d::set_error(A3D_EFORMAT, "int64_list: empty packed value");
            for (int i = 0; i < 6; ++i)
              if (klen == strlen(int_names[i]) && !memcmp(key.p, int_names[i], klen)) { *ints[i] = (int64_t)val; seen |= 1u << i; }
            break;
          }
        }
      }
    }
  }
  if (seen != 255) return a3d::set_error(A3D_EFORMAT, "example: missing features (mask 0x%x of 0xff)", seen);
  return A3D_OK;
}

int a3d_decode_raw_plus_half(const uint8_t* src, size_t bytes, float* dst) {
  if (!src || !dst || bytes % 4) return a3d::set_error(A3D_EINVAL, "decode_raw: byte count %zu not a multiple of 4", bytes);
  const size_t n = bytes / 4;
  for (size_t i = 0; i < n; ++i) {
    float v;
    memcpy(&v, src + 4 * i, 4);      // little-endian host
    dst[i] = v + 0.5f;
  }
  return A3D_OK;
}

// One pass over a framed record: payload CRC32C, Example parse and `decode_raw + 0.5` into the caller's buffers.
// The CRC is chained over [payload start, image) | image | (image, depth) | depth | (depth, payload end) and the two
// big features are decoded while their bytes are hot, so a 4.9 MB record crosses the memory bus once instead of twice.
int a3d_record_decode(const uint8_t* frame, size_t len, int verify_crc, float* image_dst, size_t image_floats,
                      float* depth_dst, size_t depth_floats, a3d_example_view* view) {
  size_t off, plen, used;
  int rc = a3d_tfrecord_next(frame, len, 0, &off, &plen, &used);
  if (rc != A3D_OK) return rc;
  const uint8_t* payload = frame + off;
  a3d_example_view ev;
  rc = a3d_example_parse(payload, plen, &ev);
  if (rc != A3D_OK) return rc;
  if (ev.image_bytes != image_floats * 4 || ev.depth_bytes != depth_floats * 4)
    return a3d::set_error(A3D_EINVAL, "record_decode: record holds %zu / %zu feature bytes, destination %zu / %zu floats",
                          ev.image_bytes, ev.depth_bytes, image_floats, depth_floats);
  if (view) *view = ev;
  static const bool hw = have_sse42();
  struct Seg { const uint8_t* p; size_t n; float* dst; };
  const bool image_first = ev.image < ev.depth;
  const Seg first = image_first ? Seg{ev.image, ev.image_bytes, image_dst} : Seg{ev.depth, ev.depth_bytes, depth_dst};
  const Seg second = image_first ? Seg{ev.depth, ev.depth_bytes, depth_dst} : Seg{ev.image, ev.image_bytes, image_dst};
  const uint8_t* cursor = payload;
  uint32_t crc = 0;
  auto crc_range = [&](const uint8_t* p, size_t n) {
    if (verify_crc && n) crc = hw ? crc32c_hw(p, n, crc) : crc32c_sw(p, n, crc);
  };
  for (const Seg& s : {first, second}) {
    crc_range(cursor, (size_t)(s.p - cursor));
    // decode in 72 KB blocks (three 3-chain CRC rounds): CRC the block, then convert it while it is in cache
    for (size_t done = 0; done < s.n;) {
      const size_t blk = std::min<size_t>(s.n - done, 73728);
      crc_range(s.p + done, blk);
      const size_t nf = blk / 4;
      float* d = s.dst + done / 4;
      for (size_t i = 0; i < nf; ++i) {
        float v;
        memcpy(&v, s.p + done + 4 * i, 4);
        d[i] = v + 0.5f;
      }
      done += blk;
    }
    cursor = s.p + s.n;
  }
  crc_range(cursor, (size_t)(payload + plen - cursor));
  if (verify_crc) {
    uint32_t want;
    memcpy(&want, payload + plen, 4);
    if (mask_crc(crc) != want) return a3d::set_error(A3D_EFORMAT, "tfrecord: corrupt payload");
  }
  return A3D_OK;
}

int64_t a3d_example_write(const float* image, int ih, int iw, int ic, const float* depth, int dh, int dw, int dc,
                          uint8_t* dst, size_t cap) {
  if (!image || !depth || ih <= 0 || iw <= 0 || ic <= 0 || dh <= 0 || dw <= 0 || dc <= 0)
    return a3d::set_error(A3D_EINVAL, "example_write: bad arguments");
  struct Feat { const char* name; int kind; uint64_t ival; const void* data; size_t bytes; };
  const size_t ib = (size_t)ih * iw * ic * 4, db = (size_t)dh * dw * dc * 4;
  // sorted by key, like the oracle's writer (protobuf leaves map order unspecified)
  const Feat feats[8] = {
      {"depth", 1, 0, depth, db}, {"depth_channels", 3, (uint64_t)dc, nullptr, 0},
      {"depth_height", 3, (uint64_t)dh, nullptr, 0}, {"depth_width", 3, (uint64_t)dw, nullptr, 0},
      {"image", 1, 0, image, ib}, {"image_channels", 3, (uint64_t)ic, nullptr, 0},
      {"image_height", 3, (uint64_t)ih, nullptr, 0}, {"image_width", 3, (uint64_t)iw, nullptr, 0}};
  size_t feat_sz[8], entry_sz[8], features_sz = 0;
  for (int i = 0; i < 8; ++i) {
    size_t list = feats[i].kind == 1 ? ld_size(feats[i].bytes) : ld_size(varint_size(feats[i].ival));
    feat_sz[i] = ld_size(list);                                     // Feature{ kind: list }
    entry_sz[i] = ld_size(strlen(feats[i].name)) + ld_size(feat_sz[i]);
    features_sz += ld_size(entry_sz[i]);
  }
  const size_t payload = ld_size(features_sz);
  const size_t total = 16 + payload;
  if (!dst || cap < total) return (int64_t)total;
  uint8_t* p = dst + 12;
  p = put_ld_head(p, 1, features_sz);
  for (int i = 0; i < 8; ++i) {
    p = put_ld_head(p, 1, entry_sz[i]);
    const size_t kl = strlen(feats[i].name);
    p = put_ld_head(p, 1, kl);
    memcpy(p, feats[i].name, kl);
    p += kl;
    p = put_ld_head(p, 2, feat_sz[i]);
    if (feats[i].kind == 1) {
      p = put_ld_head(p, 1, ld_size(feats[i].bytes));
      p = put_ld_head(p, 1, feats[i].bytes);
      memcpy(p, feats[i].data, feats[i].bytes);
      p += feats[i].bytes;
    } else {
      p = put_ld_head(p, 3, ld_size(varint_size(feats[i].ival)));
      p = put_ld_head(p, 1, varint_size(feats[i].ival));
      p = put_varint(p, feats[i].ival);
    }
  }
  const uint64_t n = payload;
  memcpy(dst, &n, 8);
  const uint32_t hc = a3d_masked_crc32c(dst, 8);
  memcpy(dst + 8, &hc, 4);
  const uint32_t pc = a3d_masked_crc32c(dst + 12, payload);
  memcpy(dst + 12 + payload, &pc, 4);
  return (int64_t)total;
}

}  // extern "C"
