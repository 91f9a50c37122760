// ga3c_checkpoint.hpp -- the checkpoint container behind ga3c_net_save / ga3c_net_load: an uncompressed .npz (a zip archive of
// .npy members, all "stored"), the format numpy.savez writes and numpy.load reads, so that a checkpoint made through the C
// ABI opens in Python and the other way round.  The reference's tf.train.Saver keys every variable by its name
// (NetworkVP.py:62-64, :267-282); the members here carry the same names ("conv11/w:0", "conv11/w/RMSProp:0", ..., "step").
// Host code only.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace ga3c_ckpt {

struct Member {
  std::string name;                 // without the ".npy" the archive member carries
  std::string descr;                // "<f4" or "<i8"
  std::vector<int64_t> shape;       // () for a scalar
  std::vector<uint8_t> bytes;       // little-endian payload, C order
};

inline uint32_t crc32(const uint8_t* p, size_t n, uint32_t crc = 0) {
  static uint32_t table[256];
  static bool ready = false;
  if (!ready) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      table[i] = c;
    }
    ready = true;
  }
  crc = ~crc;
  for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 255] ^ (crc >> 8);
  return ~crc;
}

inline void put16(std::vector<uint8_t>& o, uint32_t v) { o.push_back(v & 255); o.push_back((v >> 8) & 255); }
inline void put32(std::vector<uint8_t>& o, uint32_t v) { put16(o, v & 0xFFFF); put16(o, v >> 16); }
inline uint32_t get16(const uint8_t* p) { return p[0] | (p[1] << 8); }
inline uint32_t get32(const uint8_t* p) { return get16(p) | (get16(p + 2) << 16); }
inline uint64_t get64(const uint8_t* p) { return (uint64_t)get32(p) | ((uint64_t)get32(p + 4) << 32); }

// the .npy header (format 1.0) of a member: magic, version, length, the dict, padded with spaces to a multiple of 64 bytes
inline std::vector<uint8_t> npy_header(const Member& m) {
  std::string dict = "{'descr': '" + m.descr + "', 'fortran_order': False, 'shape': (";
  for (size_t i = 0; i < m.shape.size(); ++i) {
    dict += std::to_string((long long)m.shape[i]);
    if (m.shape.size() == 1 || i + 1 < m.shape.size()) dict += ",";
    if (i + 1 < m.shape.size()) dict += " ";
  }
  dict += "), }";
  size_t total = 10 + dict.size() + 1;
  const size_t pad = (64 - total % 64) % 64;
  dict.append(pad, ' ');
  dict += "\n";
  std::vector<uint8_t> h = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0};
  put16(h, (uint32_t)dict.size());
  h.insert(h.end(), dict.begin(), dict.end());
  return h;
}

// Writes the archive in one piece (fixed DOS date: the same members give the same bytes).  false + *err on failure.
inline bool write_npz(const std::string& path, const std::vector<Member>& members, std::string* err) {
  std::vector<uint8_t> out, central;
  for (const Member& m : members) {
    const std::string fname = m.name + ".npy";
    std::vector<uint8_t> data = npy_header(m);
    data.insert(data.end(), m.bytes.begin(), m.bytes.end());
    if (data.size() >= 0xFFFFFFFFull || out.size() >= 0xFFFFFFFFull) { *err = "archive too large for a plain zip"; return false; }
    const uint32_t crc = crc32(data.data(), data.size()), off = (uint32_t)out.size(), size = (uint32_t)data.size();
    put32(out, 0x04034b50); put16(out, 20); put16(out, 0); put16(out, 0);     // local header: version 2.0, no flags, stored
    put16(out, 0); put16(out, 0x21);                                           // time 00:00:00, date 1980-01-01
    put32(out, crc); put32(out, size); put32(out, size);
    put16(out, (uint32_t)fname.size()); put16(out, 0);
    out.insert(out.end(), fname.begin(), fname.end());
    out.insert(out.end(), data.begin(), data.end());
    put32(central, 0x02014b50); put16(central, 20); put16(central, 20); put16(central, 0); put16(central, 0);
    put16(central, 0); put16(central, 0x21);
    put32(central, crc); put32(central, size); put32(central, size);
    put16(central, (uint32_t)fname.size()); put16(central, 0); put16(central, 0); put16(central, 0); put16(central, 0);
    put32(central, 0); put32(central, off);
    central.insert(central.end(), fname.begin(), fname.end());
  }
  const uint32_t cd_off = (uint32_t)out.size(), cd_size = (uint32_t)central.size();
  out.insert(out.end(), central.begin(), central.end());
  put32(out, 0x06054b50); put16(out, 0); put16(out, 0);
  put16(out, (uint32_t)members.size()); put16(out, (uint32_t)members.size());
  put32(out, cd_size); put32(out, cd_off); put16(out, 0);
  const std::string tmp = path + ".tmp";
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) { *err = "cannot create " + tmp; return false; }
  const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
  if (fclose(f) != 0 || !ok) { remove(tmp.c_str()); *err = "short write to " + tmp; return false; }
  if (rename(tmp.c_str(), path.c_str()) != 0) { remove(tmp.c_str()); *err = "cannot rename " + tmp; return false; }
  return true;
}

// Reads every stored member of an .npz (numpy.savez's or write_npz's).  Compressed members are refused.
inline bool read_npz(const std::string& path, std::map<std::string, Member>* members, std::string* err) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) { *err = "cannot open " + path; return false; }
  std::vector<uint8_t> buf;
  uint8_t chunk[1 << 16];
  size_t got;
  while ((got = fread(chunk, 1, sizeof chunk, f)) > 0) buf.insert(buf.end(), chunk, chunk + got);
  fclose(f);
  if (buf.size() < 22) { *err = path + " is not a zip archive"; return false; }
  size_t eocd = buf.size() - 22;
  while (get32(&buf[eocd]) != 0x06054b50) {
    if (eocd == 0 || buf.size() - eocd > 66000) { *err = path + ": no end-of-central-directory record"; return false; }
    --eocd;
  }
  uint64_t count = get16(&buf[eocd + 10]), cd_off = get32(&buf[eocd + 16]);
  if (cd_off == 0xFFFFFFFFu || count == 0xFFFFu) {                       // zip64: the locator sits 20 bytes in front
    if (eocd < 20 || get32(&buf[eocd - 20]) != 0x07064b50) { *err = path + ": zip64 locator missing"; return false; }
    const uint64_t e64 = get64(&buf[eocd - 20 + 8]);
    if (e64 + 56 > buf.size() || get32(&buf[e64]) != 0x06064b50) { *err = path + ": bad zip64 record"; return false; }
    count = get64(&buf[e64 + 32]);
    cd_off = get64(&buf[e64 + 48]);
  }
  size_t p = (size_t)cd_off;
  for (uint64_t i = 0; i < count; ++i) {
    if (p + 46 > buf.size() || get32(&buf[p]) != 0x02014b50) { *err = path + ": bad central directory"; return false; }
    const uint32_t method = get16(&buf[p + 10]), nlen = get16(&buf[p + 28]), xlen = get16(&buf[p + 30]), clen = get16(&buf[p + 32]);
    uint64_t csize = get32(&buf[p + 20]), usize = get32(&buf[p + 24]), lho = get32(&buf[p + 42]);
    std::string fname(reinterpret_cast<const char*>(&buf[p + 46]), nlen);
    size_t x = p + 46 + nlen;
    const size_t xend = x + xlen;
    while (x + 4 <= xend) {                                            // zip64 extra field: the sizes that did not fit
      const uint32_t id = get16(&buf[x]), len = get16(&buf[x + 2]);
      if (id == 1) {
        size_t q = x + 4;
        if (usize == 0xFFFFFFFFu) { usize = get64(&buf[q]); q += 8; }
        if (csize == 0xFFFFFFFFu) { csize = get64(&buf[q]); q += 8; }
        if (lho == 0xFFFFFFFFu) { lho = get64(&buf[q]); q += 8; }
      }
      x += 4 + len;
    }
    p = xend + clen;
    if (method != 0) { *err = path + ": member " + fname + " is compressed (numpy.savez_compressed): not supported"; return false; }
    if (lho + 30 > buf.size() || get32(&buf[lho]) != 0x04034b50) { *err = path + ": bad local header of " + fname; return false; }
    const size_t data = (size_t)lho + 30 + get16(&buf[lho + 26]) + get16(&buf[lho + 28]);
    if (data + usize > buf.size() || usize < 10) { *err = path + ": member " + fname + " runs past the end"; return false; }
    const uint8_t* d = &buf[data];
    if (memcmp(d, "\x93NUMPY", 6) != 0) continue;                        // not an array member
    size_t hlen, hoff;
    if (d[6] == 1) { hlen = get16(d + 8); hoff = 10; } else { hlen = get32(d + 8); hoff = 12; }
    if (hoff + hlen > usize) { *err = path + ": bad .npy header in " + fname; return false; }
    const std::string dict(reinterpret_cast<const char*>(d + hoff), hlen);
    Member m;
    m.name = fname.size() > 4 && fname.compare(fname.size() - 4, 4, ".npy") == 0 ? fname.substr(0, fname.size() - 4) : fname;
    const size_t dp = dict.find("'descr':");
    const size_t q0 = dp == std::string::npos ? dp : dict.find('\'', dp + 8);
    const size_t q1 = q0 == std::string::npos ? q0 : dict.find('\'', q0 + 1);
    if (q1 == std::string::npos) { *err = path + ": no descr in " + fname; return false; }
    m.descr = dict.substr(q0 + 1, q1 - q0 - 1);
    if (dict.find("'fortran_order': False") == std::string::npos) { *err = path + ": " + fname + " is not in C order"; return false; }
    const size_t sp = dict.find("'shape':");
    const size_t b0 = sp == std::string::npos ? sp : dict.find('(', sp), b1 = b0 == std::string::npos ? b0 : dict.find(')', b0);
    if (b1 == std::string::npos) { *err = path + ": no shape in " + fname; return false; }
    int64_t cur = -1;
    for (size_t k = b0 + 1; k <= b1; ++k) {
      const char c = dict[k];
      if (c >= '0' && c <= '9') cur = (cur < 0 ? 0 : cur) * 10 + (c - '0');
      else if (cur >= 0) { m.shape.push_back(cur); cur = -1; }
    }
    m.bytes.assign(d + hoff + hlen, d + usize);
    (*members)[m.name] = std::move(m);
  }
  return true;
}

}  // namespace ga3c_ckpt
