// mat5_reader.cpp -- see mat5_reader.h.  Format reference: "MAT-File Format" (MathWorks), Level 5.
#include "mat5_reader.h"

#include <zlib.h>

#include <cstdint>
#include <cstring>
#include <fstream>

namespace {
enum { miINT8 = 1, miUINT8 = 2, miINT16 = 3, miUINT16 = 4, miINT32 = 5, miUINT32 = 6, miSINGLE = 7, miDOUBLE = 9,
       miINT64 = 12, miUINT64 = 13, miMATRIX = 14, miCOMPRESSED = 15 };
enum { mxCELL = 1, mxSTRUCT = 2 };

struct Span { const uint8_t *p; size_t n; };

struct Element { uint32_t type; Span data; size_t total; }; // total = bytes consumed including tag and padding

bool read_element(Span s, Element &e)
{
    if (s.n < 8) return false;
    uint32_t w0, w1;
    memcpy(&w0, s.p, 4);
    memcpy(&w1, s.p + 4, 4);
    if (w0 >> 16) { // small data element: 2-byte type, 2-byte size, 4 bytes of data
        e.type = w0 & 0xffff;
        const uint32_t nb = w0 >> 16;
        if (nb > 4) return false;
        e.data = Span{s.p + 4, nb};
        e.total = 8;
        return true;
    }
    e.type = w0;
    if ((size_t)w1 > s.n - 8) return false;
    e.data = Span{s.p + 8, w1};
    e.total = 8 + (size_t)w1;
    if (e.type != miCOMPRESSED) e.total = (e.total + 7) & ~(size_t)7; // padded to 8 bytes (compressed ones are not)
    if (e.total > s.n) e.total = s.n;
    return true;
}

bool inflate_all(Span in, std::vector<uint8_t> &out)
{
    z_stream z;
    memset(&z, 0, sizeof(z));
    if (inflateInit(&z) != Z_OK) return false;
    out.resize(in.n * 4 + 1024);
    z.next_in = const_cast<Bytef *>(in.p);
    z.avail_in = (uInt)in.n;
    size_t have = 0;
    int rc;
    do {
        if (have == out.size()) out.resize(out.size() * 2);
        z.next_out = out.data() + have;
        z.avail_out = (uInt)(out.size() - have);
        rc = inflate(&z, Z_NO_FLUSH);
        have = out.size() - z.avail_out;
    } while (rc == Z_OK);
    inflateEnd(&z);
    out.resize(have);
    return rc == Z_STREAM_END;
}

template <typename T>
void append_as_float(Span d, std::vector<float> &out)
{
    const size_t n = d.n / sizeof(T);
    for (size_t i = 0; i < n; i++) {
        T v;
        memcpy(&v, d.p + i * sizeof(T), sizeof(T));
        out.push_back((float)v);
    }
}

bool numeric_to_float(const Element &e, std::vector<float> &out)
{
    switch (e.type) {
    case miSINGLE: append_as_float<float>(e.data, out); return true;
    case miDOUBLE: append_as_float<double>(e.data, out); return true;
    case miINT8: append_as_float<int8_t>(e.data, out); return true;
    case miUINT8: append_as_float<uint8_t>(e.data, out); return true;
    case miINT16: append_as_float<int16_t>(e.data, out); return true;
    case miUINT16: append_as_float<uint16_t>(e.data, out); return true;
    case miINT32: append_as_float<int32_t>(e.data, out); return true;
    case miUINT32: append_as_float<uint32_t>(e.data, out); return true;
    case miINT64: append_as_float<int64_t>(e.data, out); return true;
    case miUINT64: append_as_float<uint64_t>(e.data, out); return true;
    default: return false;
    }
}

struct Matrix { uint32_t cls; std::string name; size_t numel; Span rest; }; // rest = sub-elements after the name

bool parse_matrix_header(Span body, Matrix &m)
{
    Element flags, dims, name;
    if (!read_element(body, flags) || flags.data.n < 8) return false;
    uint32_t f0;
    memcpy(&f0, flags.data.p, 4);
    m.cls = f0 & 0xff;
    Span s{body.p + flags.total, body.n - flags.total};
    if (!read_element(s, dims)) return false;
    m.numel = 1;
    for (size_t i = 0; i + 4 <= dims.data.n; i += 4) {
        int32_t d;
        memcpy(&d, dims.data.p + i, 4);
        m.numel *= (size_t)(d < 0 ? 0 : d);
    }
    s = Span{s.p + dims.total, s.n - dims.total};
    if (!read_element(s, name)) return false;
    m.name.assign((const char *)name.data.p, name.data.n);
    m.rest = Span{s.p + name.total, s.n - name.total};
    return true;
}

// the numeric field `field` of a 1x1 struct (first struct element if larger); found=false if the field is absent
bool struct_field_to_float(const Matrix &st, const char *field, std::vector<float> &out, bool &found)
{
    found = false;
    Element len_e, names_e;
    if (!read_element(st.rest, len_e) || len_e.data.n < 4) return false;
    int32_t flen;
    memcpy(&flen, len_e.data.p, 4);
    Span s{st.rest.p + len_e.total, st.rest.n - len_e.total};
    if (!read_element(s, names_e) || flen <= 0) return false;
    const size_t nfields = names_e.data.n / (size_t)flen;
    s = Span{s.p + names_e.total, s.n - names_e.total};
    for (size_t k = 0; k < nfields; k++) {
        Element fe;
        if (!read_element(s, fe)) return false;
        const char *fname = (const char *)names_e.data.p + k * flen;
        if (strncmp(fname, field, (size_t)flen) == 0 && strlen(field) < (size_t)flen + 1 && fe.type == miMATRIX) {
            Matrix fm;
            if (!parse_matrix_header(fe.data, fm)) return false;
            Element pr;
            if (fm.numel > 0) {
                if (!read_element(fm.rest, pr) || !numeric_to_float(pr, out)) return false;
                if (out.size() > fm.numel) out.resize(fm.numel);
            }
            found = true;
            return true;
        }
        s = Span{s.p + fe.total, s.n - fe.total};
    }
    return true;
}
} // namespace

bool mat5_read_lidar_scans(const std::string &path, std::vector<std::vector<float>> &scans, std::string &err,
                           const char *var_name, const char *field_name)
{
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) { err = "cannot open " + path; return false; }
    std::vector<uint8_t> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (file.size() < 128) { err = "not a MAT-file (shorter than its header)"; return false; }
    if (file[126] == 'M' && file[127] == 'I') { err = "big-endian MAT-file: only little-endian Level-5 MAT-files are supported"; return false; }
    if (!(file[126] == 'I' && file[127] == 'M')) { err = "not a Level-5 MAT-file (no endian indicator in the header)"; return false; }
    // version field 0x0200 (and the text "MATLAB 7.3 MAT-file"): an HDF5 container, not the Level-5 element stream
    if ((file[124] == 0x00 && file[125] == 0x02) || memcmp(file.data(), "MATLAB 7.3", 10) == 0) {
        err = "MATLAB v7.3 (HDF5) MAT-file: not supported, re-save with -v7 or convert with tools/mat2bin.py";
        return false;
    }
    Span s{file.data() + 128, file.size() - 128};
    std::vector<uint8_t> inflated;
    while (s.n >= 8) {
        Element e;
        if (!read_element(s, e)) { err = "corrupt data element"; return false; }
        Span body = e.data;
        uint32_t type = e.type;
        if (type == miCOMPRESSED) {
            if (!inflate_all(e.data, inflated)) { err = "zlib inflate failed"; return false; }
            Element inner;
            if (!read_element(Span{inflated.data(), inflated.size()}, inner)) { err = "corrupt compressed element"; return false; }
            body = inner.data;
            type = inner.type;
        }
        if (type == miMATRIX) {
            Matrix m;
            if (!parse_matrix_header(body, m)) { err = "corrupt matrix header"; return false; }
            if (m.name == var_name) {
                if (m.cls != mxCELL) { err = std::string("variable '") + var_name + "' is not a cell array"; return false; }
                Span c = m.rest;
                for (size_t i = 0; i < m.numel; i++) { // mxGetCell(pList, i) -> mxGetField(cell, 0, "scan")
                    Element ce;
                    if (!read_element(c, ce) || ce.type != miMATRIX) { err = "corrupt cell"; return false; }
                    Matrix cm;
                    if (!parse_matrix_header(ce.data, cm)) { err = "corrupt cell header"; return false; }
                    if (cm.cls == mxSTRUCT) {
                        std::vector<float> v;
                        bool found = false;
                        if (!struct_field_to_float(cm, field_name, v, found)) { err = "corrupt struct in cell"; return false; }
                        if (found) scans.push_back(std::move(v));
                    }
                    c = Span{c.p + ce.total, c.n - ce.total};
                }
                return true;
            }
        }
        s = Span{s.p + e.total, s.n - e.total};
    }
    err = std::string("file does not contain '") + var_name + "'";
    return false;
}
