// MRPT-shaped facade, part 1 of 3 (<mrpt/base.h>): what src/pf_localization_main.cpp (the reference's copy of MRPT's
// pf-localization application) uses of MRPT's base library - configuration files, streams,
// poses, small matrices, timing, file-system helpers, the smart-pointer idiom.
//
// MRPT is an un-vendored dependency of the reference and is absent here, so this is NOT MRPT: it is the minimal
// surface that lets that file compile unchanged (tests/test_ref_callers_compile.py) and run with the particle filter
// on the GPU behind it (lsa_pf_* / lsa_score_poses_std / lsa_field_* of include/lsa_core.h).  Two input formats differ
// by necessity (MRPT's binary serialisation cannot be read without MRPT): a ".rawlog" and a ".simplemap" are CARMEN
// logs here (ROBOTLASER1 / FLASER lines), a ".gridmap" is a binary PGM with a geometry comment (saveAsBitmapFile's
// output).  Arithmetic parity with MRPT is unpinned; see DESIGN.md §6.
#ifndef LSA_MRPT_FACADE_BASE_H
#define LSA_MRPT_FACADE_BASE_H

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#define MRPT_HAS_WXWIDGETS 0
#define MRPT_TODO(x)
#define THROW_EXCEPTION(msg) throw std::runtime_error(std::string(msg))
#define THROW_EXCEPTION_CUSTOM_MSG1(fmt, a) throw std::runtime_error(mrpt::format(fmt, a))
#define ASSERT_(c) { if (!(c)) throw std::runtime_error(std::string("Assert condition failed: ") + #c); }
#define ASSERT_FILE_EXISTS_(f) { if (!mrpt::system::fileExists(f)) throw std::runtime_error(std::string("Assert file existence failed: ") + std::string(f)); }
#define IS_CLASS(ptr, class_name) (mrpt::utils::ptr_is_class<class_name>(ptr))

namespace mrpt {

std::string format(const char* fmt, ...);

namespace system {
typedef uint64_t TTimeStamp;                               // 100 ns ticks, as MRPT's
std::string MRPT_getVersion();
std::string MRPT_getCompilationDate();
bool fileExists(const std::string& path);                  // files and directories
std::string lowerCase(const std::string& s);
std::string extractFileExtension(const std::string& path, bool ignore_gz = false);
bool createDirectory(const std::string& path);
bool deleteFiles(const std::string& wildcard);             // "<dir>/*.*"
void sleep(int ms);
void pause();
std::string dateTimeLocalToString(TTimeStamp t);
double timestampTotime_t(TTimeStamp t);
bool vectorToTextFile(const std::vector<float>& v, const std::string& file);
namespace os {
bool kbhit();
int getch();
}
}  // namespace system

namespace utils {

using mrpt::format;
typedef std::vector<int> vector_int;
typedef std::vector<double> vector_double;
typedef std::vector<float> vector_float;

inline double DEG2RAD(double d) { return d * 3.14159265358979323846 / 180.0; }
inline float DEG2RAD(float d) { return d * 3.14159265358979323846f / 180.0f; }
inline double DEG2RAD(int d) { return d * 3.14159265358979323846 / 180.0; }
inline double RAD2DEG(double r) { return r * 180.0 / 3.14159265358979323846; }

template <class T>
std::ostream& operator<<(std::ostream& o, const std::vector<T>& v)
{
    o << "[";
    for (size_t i = 0; i < v.size(); ++i) o << (i ? " " : "") << v[i];
    return o << "]";
}

// MRPT's reference-counted pointer idiom (CSomethingPtr): ->, pointer(), bool tests, up- and down-casts
class CObject {
public:
    virtual ~CObject() {}
};
template <class T>
class ptr_t {
public:
    std::shared_ptr<T> p;
    ptr_t() {}
    explicit ptr_t(T* raw) : p(raw) {}
    template <class U>
    ptr_t(const ptr_t<U>& o) : p(std::dynamic_pointer_cast<T>(o.p)) {}
    T* operator->() const { return p.get(); }
    T& operator*() const { return *p; }
    T* pointer() const { return p.get(); }
    bool present() const { return (bool)p; }
    explicit operator bool() const { return (bool)p; }
    bool operator!() const { return !p; }
    void clear() { p.reset(); }
    void clear_unique() { p.reset(); }
};
template <class C, class U>
bool ptr_is_class(const ptr_t<U>& q) { return dynamic_cast<C*>(q.pointer()) != nullptr; }

class CTicTac {
public:
    CTicTac() { Tic(); }
    void Tic() { t0 = std::chrono::steady_clock::now(); }
    double Tac() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
private:
    std::chrono::steady_clock::time_point t0;
};

// ".ini" files: [section], key = value, comments after // or ; or # at the start of a line
class CConfigFileBase {
public:
    virtual ~CConfigFileBase() {}
    std::string read_string(const std::string& sec, const std::string& name, const std::string& def, bool fail = false) const;
    int read_int(const std::string& sec, const std::string& name, int def, bool fail = false) const;
    double read_double(const std::string& sec, const std::string& name, double def, bool fail = false) const;
    float read_float(const std::string& sec, const std::string& name, float def, bool fail = false) const;
    bool read_bool(const std::string& sec, const std::string& name, bool def, bool fail = false) const;
    template <class V>
    void read_vector(const std::string& sec, const std::string& name, const V& def, V& out, bool fail = false) const
    {
        std::string raw;
        if (!lookup(sec, name, raw, fail)) { out = def; return; }
        for (size_t i = 0; i < raw.size(); ++i) if (raw[i] == '[' || raw[i] == ']' || raw[i] == ',' || raw[i] == ';') raw[i] = ' ';
        std::istringstream is(raw);
        out.clear();
        typename V::value_type v;
        while (is >> v) out.push_back(v);
    }
    bool sectionExists(const std::string& sec) const;
protected:
    bool lookup(const std::string& sec, const std::string& name, std::string& out, bool fail) const;
    std::map<std::string, std::map<std::string, std::string> > m_data;       // lower-cased section and key
};
class CConfigFile : public CConfigFileBase {
public:
    explicit CConfigFile(const std::string& file);
};

class CSerializable : public CObject {
public:
    virtual void readFromFile(const std::string& file) { (void)file; THROW_EXCEPTION("this object cannot be read from a file"); }
    virtual void writeToFile(const std::string& file) const { (void)file; }
};

class CStream {
public:
    virtual ~CStream() {}
    const std::string& fileName() const { return m_name; }
    CStream& operator>>(CSerializable& obj) { obj.readFromFile(m_name); return *this; }
    CStream& operator<<(const CSerializable& obj) { obj.writeToFile(m_name); return *this; }
    int printf(const char* fmt, ...);
protected:
    std::string m_name;
    FILE* m_out = nullptr;
};
// input: only remembers the file (and, for logs, the reading position: see CRawlog in <mrpt/slam.h>)
class CFileGZInputStream : public CStream {
public:
    CFileGZInputStream() {}
    explicit CFileGZInputStream(const std::string& file);
    ~CFileGZInputStream();
    bool open(const std::string& file);
    // state of the CARMEN reader behind CRawlog::getActionObservationPairOrObservation
    std::ifstream* log = nullptr;
    bool have_prev = false;
    double prev_odo[3] = {0, 0, 0};
};
class CFileOutputStream : public CStream {
public:
    CFileOutputStream() {}
    explicit CFileOutputStream(const std::string& file, bool append = false);
    ~CFileOutputStream();
    bool open(const std::string& file, bool append = false);
    void close();
};
typedef CFileOutputStream CFileGZOutputStream;

struct TColorf {
    TColorf(float r = 1, float g = 1, float b = 1, float a = 1) : R(r), G(g), B(b), A(a) {}
    float R, G, B, A;
};

}  // namespace utils

namespace poses {
class CPose2D {
public:
    CPose2D(double x = 0, double y = 0, double phi = 0) : m_x(x), m_y(y), m_phi(phi) {}
    double x() const { return m_x; }
    double y() const { return m_y; }
    double phi() const { return m_phi; }
    void x(double v) { m_x = v; }
    void y(double v) { m_y = v; }
    void phi(double v) { m_phi = v; }
    CPose2D operator+(const CPose2D& b) const;             // composition, this (+) b
    CPose2D operator-(const CPose2D& b) const;             // inverse composition, this (-) b
    double distanceTo(const CPose2D& b) const { return std::sqrt((m_x - b.m_x) * (m_x - b.m_x) + (m_y - b.m_y) * (m_y - b.m_y)); }
    double norm() const { return std::sqrt(m_x * m_x + m_y * m_y); }
    std::string asString() const;
private:
    double m_x, m_y, m_phi;
};
std::ostream& operator<<(std::ostream& o, const CPose2D& p);
class CPose3D {
public:
    CPose3D() : x(0), y(0), z(0), yaw(0) {}
    CPose3D(const CPose2D& p) : x(p.x()), y(p.y()), z(0), yaw(p.phi()) {}
    double x, y, z, yaw;
};
// the pose PDFs of an action: only the mean is asked for
class CPosePDF : public utils::CObject {
public:
    CPose2D mean;
    double std_xy = 0, std_phi = 0;
    CPose2D getMeanVal() const { return mean; }
};
typedef utils::ptr_t<CPosePDF> CPosePDFPtr;
}  // namespace poses

namespace math {
using namespace mrpt::poses;
class CMatrixDouble {
public:
    CMatrixDouble(size_t r = 0, size_t c = 0) : m_r(r), m_c(c), m_v(r * c, 0.0) {}
    void setSize(size_t r, size_t c) { m_r = r; m_c = c; m_v.assign(r * c, 0.0); }
    size_t getRowCount() const { return m_r; }
    size_t getColCount() const { return m_c; }
    double& operator()(size_t i, size_t j) { return m_v[i * m_c + j]; }
    double operator()(size_t i, size_t j) const { return m_v[i * m_c + j]; }
    void loadFromTextFile(const std::string& file);       // rows of numbers, '%' and '#' comments
    double det() const;
protected:
    size_t m_r, m_c;
    std::vector<double> m_v;
};
class CMatrixDouble33 : public CMatrixDouble {
public:
    CMatrixDouble33() : CMatrixDouble(3, 3) {}
};
double round_10power(double v, int power10);
void confidenceIntervals(const std::vector<double>& data, double& mean, double& lo, double& hi, double conf = 0.1,
                         size_t n_bins = 1000);
}  // namespace math

namespace random {
class CRandomGenerator {
public:
    void randomize();
    void randomize(uint32_t seed);
    double drawUniform(double lo, double hi);
    double drawGaussian1D_normalized();
private:
    uint64_t m_state = 0x9E3779B97F4A7C15ull;
    bool m_have = false;
    double m_spare = 0;
};
extern CRandomGenerator randomGenerator;
}  // namespace random

namespace utils { using namespace mrpt::poses; }
namespace system { using mrpt::utils::vector_double; }

}  // namespace mrpt

#endif
