// Host-side ingest / egress around the path (next row N2 of the scope table).
//   pct_text_shape / pct_text_load : the "x y z [nx ny nz]" text scans PointCloud.read_from_file parses with
//                                    np.loadtxt (/root/reference/pointCloudToolbox.py:51) -- whitespace-separated
//                                    columns, '#' comments, blank lines skipped; values parsed as correctly rounded
//                                    float64 (bit-identical to Python's float()), multi-threaded over line-aligned chunks
//   pct_write_ply_ascii            : the ASCII PLY with K/H the reference writes one f-string per vertex
//                                    (/root/reference/utils.py:538-551); numbers formatted exactly as the f-string
//                                    does: repr(float(np.float32 value))
// Pure C++17 (no device code); lives in the same shared library so the ctypes shim has one ABI.
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <limits>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/pct_hip.h"

namespace {

struct Mapped {
    const char* p = nullptr;
    size_t n = 0;
    int fd = -1;
    bool open(const char* path) {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) return false;
        n = (size_t)st.st_size;
        if (n == 0) { p = ""; return true; }
        void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) return false;
        p = (const char*)m;
        return true;
    }
    ~Mapped() {
        if (p && n) munmap((void*)p, n);
        if (fd >= 0) ::close(fd);
    }
};

inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\r' || c == ',' ; }

// Clinger's fast path: a decimal with <= 15 significant digits and |10-exponent| <= 22 is mantissa * 10^e or
// mantissa / 10^-e with both operands exact in binary64, so ONE correctly rounded IEEE operation gives the
// correctly rounded result (what strtod / Python float() return).  Anything else falls through to from_chars.
inline bool fast_decimal(const char* b, const char* e, double& out) {
    static const double p10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15,
                                   1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    const char* c = b;
    bool neg = false;
    if (c < e && *c == '-') { neg = true; ++c; }
    uint64_t mant = 0;
    int nd = 0, frac = 0;
    bool any = false, seen_nz = false;
    while (c < e && *c >= '0' && *c <= '9') { mant = mant * 10 + (uint64_t)(*c - '0'); seen_nz |= *c != '0'; nd += seen_nz; any = true; ++c; if (nd > 15) return false; }
    if (c < e && *c == '.') {
        ++c;
        while (c < e && *c >= '0' && *c <= '9') { mant = mant * 10 + (uint64_t)(*c - '0'); seen_nz |= *c != '0'; nd += seen_nz; ++frac; any = true; ++c; if (nd > 15) return false; }
    }
    if (!any) return false;
    int ex = 0;
    if (c < e && (*c == 'e' || *c == 'E')) {
        ++c;
        bool eneg = false;
        if (c < e && (*c == '+' || *c == '-')) { eneg = *c == '-'; ++c; }
        if (c >= e) return false;
        int ed = 0;
        while (c < e && *c >= '0' && *c <= '9') { ex = ex * 10 + (*c - '0'); ++c; if (++ed > 4) return false; }
        if (eneg) ex = -ex;
    }
    if (c != e) return false;
    const int e10 = ex - frac;
    if (e10 < -22 || e10 > 22) return false;
    double v = (double)mant;
    v = e10 < 0 ? v / p10[-e10] : v * p10[e10];
    out = neg ? -v : v;
    return true;
}

// parses one line [b,e): appends values to out; returns number of values, -1 on a malformed token
int parse_line(const char* b, const char* e, double* out, int max_vals) {
    int n = 0;
    const char* c = b;
    while (c < e) {
        while (c < e && is_space(*c)) ++c;
        if (c >= e || *c == '#') break;
        const char* t = c;
        while (c < e && !is_space(*c) && *c != '#') ++c;
        const char* tb = t;
        if (tb < c && *tb == '+') ++tb;                       // from_chars rejects a leading '+'
        double v = 0;
        if (fast_decimal(tb, c, v)) {
            if (n < max_vals) out[n] = v;
            ++n;
            continue;
        }
        auto r = std::from_chars(tb, c, v);
        if (r.ec != std::errc() || r.ptr != c) {
            std::string tok(t, c);                           // rare forms (hex floats, ...) : strtod decides
            char* endp = nullptr;
            v = strtod(tok.c_str(), &endp);
            if (endp == tok.c_str() || *endp != 0) return -1;
        }
        if (n < max_vals) out[n] = v;
        ++n;
    }
    return n;
}

std::vector<size_t> chunk_starts(const Mapped& m, int threads) {
    std::vector<size_t> s{0};
    for (int i = 1; i < threads; ++i) {
        size_t pos = m.n / threads * i;
        while (pos < m.n && m.p[pos] != '\n') ++pos;
        if (pos < m.n) ++pos;
        if (pos > s.back() && pos < m.n) s.push_back(pos);
    }
    s.push_back(m.n);
    return s;
}

// What an f-string prints for a np.float32 (utils.py:549-551): NumPy's floating scalars format through Python's
// float, i.e. repr(float(x)) of the value widened to binary64 -- shortest digits that round-trip the double,
// positional for decimal exponents -4..15, else scientific with at least two exponent digits.
int format_py_float(double x, char* out) {
    if (std::isnan(x)) { memcpy(out, "nan", 3); return 3; }
    if (std::isinf(x)) { if (x < 0) { memcpy(out, "-inf", 4); return 4; } memcpy(out, "inf", 3); return 3; }
    char* o = out;
    if (std::signbit(x)) { *o++ = '-'; x = -x; }
    if (x == 0.0) { memcpy(o, "0.0", 3); return (int)(o - out) + 3; }
    char sci[40];
    auto r = std::to_chars(sci, sci + sizeof(sci) - 1, x, std::chars_format::scientific);   // d[.ddd]e[+-]XX, shortest
    *r.ptr = 0;                                               // to_chars does not terminate; atoi below needs it
    const char* epos = (const char*)memchr(sci, 'e', r.ptr - sci);
    char digits[24];
    int nd = 0;
    for (const char* c = sci; c < epos; ++c)
        if (*c != '.') digits[nd++] = *c;
    const int exp10 = atoi(epos + 1);
    if (exp10 >= -4 && exp10 < 16) {                          // positional
        if (exp10 >= 0) {
            for (int i = 0; i <= exp10; ++i) *o++ = i < nd ? digits[i] : '0';
            *o++ = '.';
            if (nd > exp10 + 1) for (int i = exp10 + 1; i < nd; ++i) *o++ = digits[i];
            else *o++ = '0';
        } else {
            *o++ = '0'; *o++ = '.';
            for (int i = 0; i < -exp10 - 1; ++i) *o++ = '0';
            for (int i = 0; i < nd; ++i) *o++ = digits[i];
        }
    } else {                                                  // scientific: d.ddde+XX
        *o++ = digits[0];
        if (nd > 1) { *o++ = '.'; for (int i = 1; i < nd; ++i) *o++ = digits[i]; }
        *o++ = 'e';
        *o++ = exp10 < 0 ? '-' : '+';
        const int ae = exp10 < 0 ? -exp10 : exp10;
        if (ae < 10) *o++ = '0';
        o += snprintf(o, 8, "%d", ae);
    }
    return (int)(o - out);
}

// One parsing pass: every thread parses its line-aligned chunk into a private vector; pct_text_shape keeps the
// parsed chunks in a small per-thread-of-caller cache so that the following pct_text_load only concatenates them.
struct ParsedFile {
    std::string path;
    int64_t rows = 0;
    int cols = 0;
    std::vector<std::vector<double>> parts;
};
thread_local ParsedFile g_last;

int parse_file(const char* path, ParsedFile& pf) {
    Mapped m;
    if (!m.open(path)) return PCT_ERR_INVALID;
    int threads = (int)std::thread::hardware_concurrency();
    if (threads < 1) threads = 1;
    if (threads > 32) threads = 32;
    if (m.n < (1u << 18)) threads = 1;
    auto starts = chunk_starts(m, threads);
    const int nch = (int)starts.size() - 1;
    std::vector<int> ncol(nch, 0), bad(nch, 0);
    pf.parts.assign(nch, {});
    std::vector<std::thread> th;
    for (int t = 0; t < nch; ++t)
        th.emplace_back([&, t] {
            const char* c = m.p + starts[t];
            const char* end = m.p + starts[t + 1];
            std::vector<double>& v = pf.parts[t];
            v.reserve((size_t)(end - c) / 8);
            double tmp[64];
            while (c < end) {
                const char* nl = (const char*)memchr(c, '\n', end - c);
                const char* e = nl ? nl : end;
                int nv = parse_line(c, e, tmp, 64);
                if (nv < 0 || nv > 64) bad[t] = 1;
                else if (nv > 0) {
                    if (ncol[t] == 0) ncol[t] = nv;
                    else if (ncol[t] != nv) bad[t] = 1;          // np.loadtxt: inconsistent column count
                    v.insert(v.end(), tmp, tmp + nv);
                }
                c = nl ? nl + 1 : end;
            }
        });
    for (auto& x : th) x.join();
    int nc = 0;
    int64_t total = 0;
    for (int t = 0; t < nch; ++t) {
        if (bad[t]) return PCT_ERR_INVALID;
        if (ncol[t]) { if (nc == 0) nc = ncol[t]; else if (nc != ncol[t]) return PCT_ERR_INVALID; }
        total += (int64_t)pf.parts[t].size();
    }
    pf.path = path;
    pf.cols = nc;
    pf.rows = nc ? total / nc : 0;
    return PCT_OK;
}

}  // namespace

extern "C" {

int pct_text_shape(const char* path, int64_t* rows, int32_t* cols) {
    if (!path || !rows || !cols) return PCT_ERR_INVALID;
    g_last = ParsedFile();
    const int st = parse_file(path, g_last);
    if (st != PCT_OK) { g_last = ParsedFile(); return st; }
    *rows = g_last.rows;
    *cols = g_last.cols;
    return PCT_OK;
}

int pct_text_load(const char* path, int64_t rows, int32_t cols, double* out) {
    if (!path || !out || rows < 0 || cols < 1 || cols > 64) return PCT_ERR_INVALID;
    if (g_last.path != path) {                               // not primed by pct_text_shape on this thread: parse now
        g_last = ParsedFile();
        const int st = parse_file(path, g_last);
        if (st != PCT_OK) { g_last = ParsedFile(); return st; }
    }
    if (g_last.rows != rows || g_last.cols != cols) { g_last = ParsedFile(); return PCT_ERR_INVALID; }
    std::vector<std::thread> th;
    size_t off = 0;
    for (auto& part : g_last.parts) {
        double* dst = out + off;
        const std::vector<double>* src = &part;
        th.emplace_back([dst, src] { if (!src->empty()) memcpy(dst, src->data(), src->size() * sizeof(double)); });
        off += part.size();
    }
    for (auto& x : th) x.join();
    g_last = ParsedFile();
    return PCT_OK;
}

int pct_format_float(double x, char* out32) { return out32 ? format_py_float(x, out32) : 0; }

int pct_write_ply_ascii(const char* path, const float* xyz, const float* gaussian, const float* mean, int64_t n) {
    if (!path || !xyz || !gaussian || !mean || n < 0) return PCT_ERR_INVALID;
    FILE* f = fopen(path, "wb");
    if (!f) return PCT_ERR_INVALID;
    fprintf(f, "ply\nformat ascii 1.0\nelement vertex %lld\nproperty float x\nproperty float y\nproperty float z\n"
               "property float gaussian_curvature\nproperty float mean_curvature\nend_header\n", (long long)n);   // utils.py:539-547
    int threads = (int)std::thread::hardware_concurrency();
    if (threads < 1) threads = 1;
    if (threads > 32) threads = 32;
    if (n < 100000) threads = 1;
    std::vector<std::string> parts(threads);
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t)
        th.emplace_back([&, t] {
            const int64_t lo = n * t / threads, hi = n * (t + 1) / threads;
            std::string& s = parts[t];
            s.reserve((size_t)(hi - lo) * 64);
            char buf[40];
            for (int64_t i = lo; i < hi; ++i) {
                const float v[5] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], gaussian[i], mean[i]};
                for (int j = 0; j < 5; ++j) {
                    s.append(buf, (size_t)format_py_float((double)v[j], buf));
                    s.push_back(j == 4 ? '\n' : ' ');
                }
            }
        });
    for (auto& x : th) x.join();
    bool ok = true;
    for (auto& s : parts) ok = ok && fwrite(s.data(), 1, s.size(), f) == s.size();
    ok = fclose(f) == 0 && ok;
    return ok ? PCT_OK : PCT_ERR_INVALID;
}

}  // extern "C"


// ---------------------------------------------------------------------------------------------------------------
// The three matrix norms of the constructor (pct:45-47: np.linalg.norm(points, 1 | 2 | inf); nothing reads them, but
// a drop-in constructor computes them): ONE multi-threaded pass over the (N, 3) array instead of numpy's transposed
// copies -- column sums of |x| in float64, the largest row sum of |x| in the array's own dtype and numpy's order
// ((|x| + |y|) + |z|), the six entries of the 3 x 3 Gram matrix in float64 (its largest eigenvalue is sigma_1^2).
// out9 = {sum|x|, sum|y|, sum|z|, max row sum, gxx, gxy, gxz, gyy, gyz, gzz} -> 10 doubles; non-finite input makes
// the Gram entries non-finite (the caller raises what numpy's SVD raises).
// ---------------------------------------------------------------------------------------------------------------
namespace {
template <class T>
void norms_range(const T* p, int64_t lo, int64_t hi, double* o) {
    double s0 = 0, s1 = 0, s2 = 0, g00 = 0, g01 = 0, g02 = 0, g11 = 0, g12 = 0, g22 = 0;
    T row_max = 0;
    bool nan_row = false;
    for (int64_t i = lo; i < hi; ++i) {
        const T x = p[3 * i], y = p[3 * i + 1], z = p[3 * i + 2];
        const T ax = x < 0 ? -x : x, ay = y < 0 ? -y : y, az = z < 0 ? -z : z;
        s0 += (double)ax; s1 += (double)ay; s2 += (double)az;
        const T r = (ax + ay) + az;
        if (r > row_max) row_max = r;
        nan_row |= !(r == r);
        const double dx = x, dy = y, dz = z;
        g00 += dx * dx; g01 += dx * dy; g02 += dx * dz; g11 += dy * dy; g12 += dy * dz; g22 += dz * dz;
    }
    o[0] = s0; o[1] = s1; o[2] = s2; o[3] = nan_row ? std::numeric_limits<double>::quiet_NaN() : (double)row_max;
    o[4] = g00; o[5] = g01; o[6] = g02; o[7] = g11; o[8] = g12; o[9] = g22;
}

template <class T>
int matrix_norms(const T* p, int64_t n, double* out10) {
    if (!p || !out10 || n <= 0) return 1;
    int threads = (int)std::thread::hardware_concurrency();
    if (threads < 1) threads = 1;
    if (threads > 16) threads = 16;
    if (n < 200000) threads = 1;
    std::vector<double> part((size_t)threads * 10, 0.0);
    std::vector<std::thread> th;
    const int64_t per = (n + threads - 1) / threads;
    for (int t = 1; t < threads; ++t)
        th.emplace_back([&, t] { norms_range(p, std::min<int64_t>(n, t * per), std::min<int64_t>(n, (t + 1) * per), &part[(size_t)t * 10]); });
    norms_range(p, 0, std::min<int64_t>(n, per), &part[0]);
    for (auto& x : th) x.join();
    for (int j = 0; j < 10; ++j) out10[j] = 0;
    out10[3] = -1;
    for (int t = 0; t < threads; ++t) {
        const double* o = &part[(size_t)t * 10];
        if ((int64_t)t * per >= n) break;
        for (int j = 0; j < 10; ++j) {
            if (j == 3) { if (!(o[3] <= out10[3])) out10[3] = o[3] != o[3] ? o[3] : (o[3] > out10[3] ? o[3] : out10[3]); }
            else out10[j] += o[j];
        }
        if (out10[3] != out10[3]) {}
    }
    return 0;
}
}  // namespace

extern "C" int pct_matrix_norms_f32(const float* xyz, int64_t n, double* out10) { return matrix_norms(xyz, n, out10); }
extern "C" int pct_matrix_norms_f64(const double* xyz, int64_t n, double* out10) { return matrix_norms(xyz, n, out10); }
