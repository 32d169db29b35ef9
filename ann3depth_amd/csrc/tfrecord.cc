// tfrecord.cc — host side of the dataset plugin: CRC32C, TFRecord framing, and a hand-written protobuf wire
// parser/writer for the 8-feature tf.train.Example that ann3depth stores (reference: src/data.py:62-86 reads it,
// tools/data_tf_converter.py:27-53 writes it).  No TensorFlow, no libprotobuf.
#include <algorithm>
#include <cstring>
#include <initializer_list>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

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

// The converter writes png_u8 / 255 - 0.5 in float32 (tools/data_tf_converter.py:36-37 of the reference), so a record's
// floats are normally 256 distinct values.  quantise() recovers k and checks, bit for bit, that the float IS
// fl(fl(k / 255) - 0.5): returns false at the first float that is not (the caller then decodes the feature as float32).
#if defined(__x86_64__)
// eight floats per round on AVX2 (the scalar loop below costs a reader thread more than the record's CRC)
__attribute__((target("avx2"))) static size_t quantise_avx2(const uint8_t* src, size_t n, uint8_t* dst, bool* ok) {
  const __m256 half = _mm256_set1_ps(0.5f), scale = _mm256_set1_ps(255.0f), zero = _mm256_setzero_ps(),
               top = _mm256_set1_ps(256.0f);
  __m256 bad = _mm256_setzero_ps();
  size_t i = 0;
  for (; i + 16 <= n; i += 16) {
    __m256i k[2];
    for (int h = 0; h < 2; ++h) {
      const __m256 v = _mm256_loadu_ps(reinterpret_cast<const float*>(src + 4 * (i + 8 * h)));
      const __m256 kf = _mm256_add_ps(_mm256_mul_ps(_mm256_add_ps(v, half), scale), half);
      const __m256 in = _mm256_and_ps(_mm256_cmp_ps(kf, zero, _CMP_GE_OQ), _mm256_cmp_ps(kf, top, _CMP_LT_OQ));
      k[h] = _mm256_cvttps_epi32(_mm256_and_ps(kf, in));
      const __m256 back = _mm256_sub_ps(_mm256_div_ps(_mm256_cvtepi32_ps(k[h]), scale), half);
      bad = _mm256_or_ps(bad, _mm256_or_ps(_mm256_cmp_ps(back, v, _CMP_NEQ_UQ), _mm256_cmp_ps(in, zero, _CMP_EQ_OQ)));
    }
    // 16 x int32 -> 16 x uint8 in order
    const __m256i p16 = _mm256_permute4x64_epi64(_mm256_packus_epi32(k[0], k[1]), 0xD8);
    const __m128i p8 = _mm_packus_epi16(_mm256_castsi256_si128(p16), _mm256_extracti128_si256(p16, 1));
    _mm_storeu_si128(reinterpret_cast<__m128i*>(dst + i), p8);
  }
  *ok = _mm256_movemask_ps(bad) == 0;
  return i;
}
#endif

static bool quantise(const uint8_t* src, size_t n, uint8_t* dst) {
  uint32_t bad = 0;
  size_t i = 0;
#if defined(__x86_64__)
  static const bool avx2 = __builtin_cpu_supports("avx2");
  if (avx2) {
    bool ok = true;
    i = quantise_avx2(src, n, dst, &ok);
    if (!ok) return false;
  }
#endif
  for (; i < n; ++i) {
    float v;
    memcpy(&v, src + 4 * i, 4);
    const float kf = (v + 0.5f) * 255.0f + 0.5f;               // in [0.5, 255.5] when representable; NaN fails `in`
    const bool in = kf >= 0.f && kf < 256.f;
    const int k = (int)(in ? kf : 0.f);
    const float back = (float)k / 255.0f - 0.5f;
    bad |= (uint32_t)(!(back == v)) | (uint32_t)(!in);
    dst[i] = (uint8_t)k;
  }
  return bad == 0;
}

// a3d_record_decode that ships a feature as uint8 when every one of its floats has the converter's form k/255 - 0.5
// (4x fewer bytes to pin, to DMA and to read back on the device, where a3d_resize_bilinear_tf1_ex rebuilds
// fl(fl(fl(k/255) - 0.5) + 0.5) — the value a3d_record_decode would have stored — bit for bit).  A feature that holds
// any other float is decoded as float32 + 0.5 exactly like a3d_record_decode.  *kinds: bit 0 = image went to image_u8,
// bit 1 = depth went to depth_u8; the other destination of a feature is left untouched.
int a3d_record_decode_u8(const uint8_t* frame, size_t len, int verify_crc, uint8_t* image_u8, float* image_f32,
                         size_t image_count, uint8_t* depth_u8, float* depth_f32, size_t depth_count,
                         a3d_example_view* view, int* kinds) {
  if (!kinds || !image_u8 || !image_f32 || !depth_u8 || !depth_f32)
    return a3d::set_error(A3D_EINVAL, "record_decode_u8: null argument");
  size_t off, plen, used;
  int rc = a3d_tfrecord_next(frame, len, 0, &off, &plen, &used);
  if (rc != A3D_OK) return rc;
  const uint8_t* payload = frame + off;
  a3d_example_view ev;
  rc = a3d_example_parse(payload, plen, &ev);
  if (rc != A3D_OK) return rc;
  if (ev.image_bytes != image_count * 4 || ev.depth_bytes != depth_count * 4)
    return a3d::set_error(A3D_EINVAL, "record_decode_u8: record holds %zu / %zu feature bytes, destination %zu / %zu floats",
                          ev.image_bytes, ev.depth_bytes, image_count, depth_count);
  if (view) *view = ev;
  static const bool hw = have_sse42();
  struct Seg { const uint8_t* p; size_t n; uint8_t* q; float* f; int bit; };
  const bool image_first = ev.image < ev.depth;
  const Seg img{ev.image, ev.image_bytes, image_u8, image_f32, 1}, dep{ev.depth, ev.depth_bytes, depth_u8, depth_f32, 2};
  const Seg first = image_first ? img : dep, second = image_first ? dep : img;
  const uint8_t* cursor = payload;
  uint32_t crc = 0;
  auto crc_range = [&](const uint8_t* p, size_t n) {
    if (verify_crc && n) crc = hw ? crc32c_hw(p, n, crc) : crc32c_sw(p, n, crc);
  };
  *kinds = 0;
  for (const Seg& s : {first, second}) {
    crc_range(cursor, (size_t)(s.p - cursor));
    bool as_u8 = true;
    for (size_t done = 0; done < s.n;) {                     // 72 KB blocks: CRC a block, then convert it while it is in cache
      const size_t blk = std::min<size_t>(s.n - done, 73728);
      crc_range(s.p + done, blk);
      if (as_u8 && !quantise(s.p + done, blk / 4, s.q + done / 4)) {
        as_u8 = false;                                        // not the converter's floats: this feature goes out as float32
        for (size_t i = 0; i < done / 4; ++i) {               // (what was already quantised is decoded again)
          float v;
          memcpy(&v, s.p + 4 * i, 4);
          s.f[i] = v + 0.5f;
        }
      }
      if (!as_u8) {
        float* d = s.f + done / 4;
        for (size_t i = 0; i < blk / 4; ++i) {
          float v;
          memcpy(&v, s.p + done + 4 * i, 4);
          d[i] = v + 0.5f;
        }
      }
      done += blk;
    }
    if (as_u8) *kinds |= s.bit;
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

// A reader thread's unit of work: n framed records into n slots of the staging pool in ONE call (the thread holds the
// host language's interpreter lock only between calls: with one record per call sixteen readers took it from the thread
// that launches the training step often enough to cost 10 % of the loop, tools/bench_input.py).  u8 pools NULL: plain
// float32 decode (a3d_record_decode); otherwise a3d_record_decode_u8, kinds[i] as there.  Every record must have the
// sizes dims = {image h, w, c, depth h, w, c}: records of different sizes cannot share a batch.
int a3d_records_decode(const void* const* frames, const size_t* lens, int n, int verify_crc, const int64_t* dims,
                       uint8_t* image_u8_pool, float* image_f32_pool, uint8_t* depth_u8_pool, float* depth_f32_pool,
                       const int32_t* slots, int nslots, int32_t* kinds) {
  if (!frames || !lens || !dims || !image_f32_pool || !depth_f32_pool || !slots || !kinds || n <= 0 || nslots <= 0)
    return a3d::set_error(A3D_EINVAL, "records_decode: bad arguments");
  size_t cnt[2] = {1, 1};
  for (int j = 0; j < 6; ++j) {        // positive, and the element counts (times the slot count) stay far inside size_t
    if (dims[j] <= 0 || dims[j] > (int64_t)1 << 31) return a3d::set_error(A3D_EINVAL, "records_decode: dims[%d] = %lld", j, (long long)dims[j]);
    if (cnt[j / 3] > ((size_t)1 << 40) / (size_t)dims[j]) return a3d::set_error(A3D_EINVAL, "records_decode: feature too large");
    cnt[j / 3] *= (size_t)dims[j];
  }
  const size_t ic = cnt[0], dc = cnt[1];
  for (int i = 0; i < n; ++i)
    if (slots[i] < 0 || slots[i] >= nslots)
      return a3d::set_error(A3D_EINVAL, "records_decode: slot %d outside the pool's %d slots", slots[i], nslots);
  for (int i = 0; i < n; ++i) {
    const size_t s = (size_t)slots[i];
    a3d_example_view ev;
    int k = 0, rc;
    if (image_u8_pool && depth_u8_pool)
      rc = a3d_record_decode_u8(static_cast<const uint8_t*>(frames[i]), lens[i], verify_crc, image_u8_pool + s * ic,
                                image_f32_pool + s * ic, ic, depth_u8_pool + s * dc, depth_f32_pool + s * dc, dc, &ev, &k);
    else
      rc = a3d_record_decode(static_cast<const uint8_t*>(frames[i]), lens[i], verify_crc, image_f32_pool + s * ic, ic,
                             depth_f32_pool + s * dc, dc, &ev);
    if (rc != A3D_OK) return rc;
    if (ev.image_height != dims[0] || ev.image_width != dims[1] || ev.image_channels != dims[2] || ev.depth_height != dims[3] ||
        ev.depth_width != dims[4] || ev.depth_channels != dims[5])
      return a3d::set_error(A3D_EINVAL, "records_decode: record is %lldx%lldx%lld / %lldx%lldx%lld, the pool holds %lldx%lldx%lld / "
                            "%lldx%lldx%lld: records of different sizes cannot be batched", (long long)ev.image_height,
                            (long long)ev.image_width, (long long)ev.image_channels, (long long)ev.depth_height,
                            (long long)ev.depth_width, (long long)ev.depth_channels, (long long)dims[0], (long long)dims[1],
                            (long long)dims[2], (long long)dims[3], (long long)dims[4], (long long)dims[5]);
    kinds[i] = k;
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
