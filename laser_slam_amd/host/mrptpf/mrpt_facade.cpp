// Bodies of the MRPT-shaped facade (mrpt/base.h, mrpt/slam.h, mrpt/gui.h): the host side of
// src/pf_localization_main.cpp over the C ABI of include/lsa_core.h.  What touches the particles or the map's
// likelihood field runs on the GPU; files, configuration and bookkeeping are plain C++.
#include "mrpt/gui.h"

#include <dirent.h>
#include <sys/stat.h>
#include <sys/types.h>

#include <ctime>
#include <random>
#include <thread>

#include "lsa_core.h"

namespace {
void lsa_ok(int rc, const char* what)
{
    if (rc != LSA_OK) throw std::runtime_error(std::string(what) + ": " + lsa_last_error());
}
std::string trim(const std::string& s)
{
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) ++a;
    while (b > a && isspace((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}
std::string lower(std::string s)
{
    for (size_t i = 0; i < s.size(); ++i) s[i] = (char)tolower((unsigned char)s[i]);
    return s;
}
// one context for everything the facade does on the device (the laser parameters play no part in these calls)
lsa_ctx* device_ctx()
{
    static lsa_ctx* c = nullptr;
    if (!c) {
        lsa_laser_param p;
        lsa_ok(lsa_param_preset(&p, "LMS211"), "lsa_param_preset");
        lsa_ok(lsa_ctx_create(&c, lsa_default_device(), &p), "lsa_ctx_create");
    }
    return c;
}
double wrap_pi(double a)
{
    while (a > 3.14159265358979323846) a -= 2 * 3.14159265358979323846;
    while (a <= -3.14159265358979323846) a += 2 * 3.14159265358979323846;
    return a;
}
}  // namespace

namespace mrpt {

std::string format(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    va_list ap2;
    va_copy(ap2, ap);
    const int n = vsnprintf(nullptr, 0, fmt, ap);
    va_end(ap);
    std::string out(n > 0 ? (size_t)n : 0, '\0');
    if (n > 0) vsnprintf(&out[0], (size_t)n + 1, fmt, ap2);
    va_end(ap2);
    return out;
}

// ---- system ---------------------------------------------------------------------------------------------------------
namespace system {
std::string MRPT_getVersion() { return std::string("facade over ") + lsa_version(); }
std::string MRPT_getCompilationDate() { return __DATE__; }
bool fileExists(const std::string& path)
{
    struct stat st;
    return !path.empty() && stat(path.c_str(), &st) == 0;
}
std::string lowerCase(const std::string& s) { return lower(s); }
std::string extractFileExtension(const std::string& path, bool ignore_gz)
{
    std::string p = path;
    for (;;) {
        const size_t dot = p.find_last_of('.'), slash = p.find_last_of('/');
        if (dot == std::string::npos || (slash != std::string::npos && dot < slash)) return "";
        const std::string ext = p.substr(dot + 1);
        if (ignore_gz && lower(ext) == "gz") { p = p.substr(0, dot); continue; }
        return ext;
    }
}
bool createDirectory(const std::string& path)
{
    if (fileExists(path)) return true;
    return mkdir(path.c_str(), 0755) == 0;
}
bool deleteFiles(const std::string& wildcard)
{
    const size_t slash = wildcard.find_last_of('/');
    const std::string dir = slash == std::string::npos ? "." : wildcard.substr(0, slash);
    DIR* d = opendir(dir.c_str());
    if (!d) return false;
    while (struct dirent* e = readdir(d)) {
        const std::string full = dir + "/" + e->d_name;
        struct stat st;
        if (stat(full.c_str(), &st) == 0 && S_ISREG(st.st_mode)) remove(full.c_str());
    }
    closedir(d);
    return true;
}
void sleep(int ms) { std::this_thread::sleep_for(std::chrono::milliseconds(ms > 0 ? ms : 0)); }
void pause() {}
std::string dateTimeLocalToString(TTimeStamp t)
{
    const time_t s = (time_t)timestampTotime_t(t);
    char buf[64];
    struct tm tmv;
    localtime_r(&s, &tmv);
    strftime(buf, sizeof buf, "%Y/%m/%d,%H:%M:%S", &tmv);
    return buf;
}
double timestampTotime_t(TTimeStamp t) { return (double)t / 1e7; }
bool vectorToTextFile(const std::vector<float>& v, const std::string& file)
{
    FILE* f = fopen(file.c_str(), "w");
    if (!f) return false;
    for (size_t i = 0; i < v.size(); ++i) fprintf(f, "%.9g ", (double)v[i]);
    fprintf(f, "\n");
    fclose(f);
    return true;
}
namespace os {
bool kbhit() { return false; }
int getch() { return 0; }
}
}  // namespace system

// ---- utils ------------------------------------------------------------------------------------------------------------
namespace utils {
CConfigFile::CConfigFile(const std::string& file)
{
    std::ifstream in(file.c_str());
    if (!in) THROW_EXCEPTION("cannot open configuration file " + file);
    std::string line, sec;
    while (std::getline(in, line)) {
        const size_t c2 = line.find("//");
        if (c2 != std::string::npos) line = line.substr(0, c2);
        line = trim(line);
        if (line.empty() || line[0] == ';' || line[0] == '#') continue;
        if (line[0] == '[') {
            const size_t e = line.find(']');
            sec = lower(trim(line.substr(1, e == std::string::npos ? std::string::npos : e - 1)));
            m_data[sec];
            continue;
        }
        const size_t eq = line.find('=');
        if (eq == std::string::npos) continue;
        m_data[sec][lower(trim(line.substr(0, eq)))] = trim(line.substr(eq + 1));
    }
}
bool CConfigFileBase::sectionExists(const std::string& sec) const { return m_data.count(lower(sec)) != 0; }
bool CConfigFileBase::lookup(const std::string& sec, const std::string& name, std::string& out, bool fail) const
{
    auto s = m_data.find(lower(sec));
    if (s != m_data.end()) {
        auto k = s->second.find(lower(name));
        if (k != s->second.end()) { out = k->second; return true; }
    }
    if (fail) THROW_EXCEPTION("Value '" + name + "' not found in section '" + sec + "' of the configuration file");
    return false;
}
std::string CConfigFileBase::read_string(const std::string& sec, const std::string& name, const std::string& def, bool fail) const
{
    std::string v;
    return lookup(sec, name, v, fail) ? v : def;
}
int CConfigFileBase::read_int(const std::string& sec, const std::string& name, int def, bool fail) const
{
    std::string v;
    return lookup(sec, name, v, fail) ? (int)strtol(v.c_str(), nullptr, 0) : def;
}
double CConfigFileBase::read_double(const std::string& sec, const std::string& name, double def, bool fail) const
{
    std::string v;
    return lookup(sec, name, v, fail) ? strtod(v.c_str(), nullptr) : def;
}
float CConfigFileBase::read_float(const std::string& sec, const std::string& name, float def, bool fail) const
{
    return (float)read_double(sec, name, def, fail);
}
bool CConfigFileBase::read_bool(const std::string& sec, const std::string& name, bool def, bool fail) const
{
    std::string v;
    if (!lookup(sec, name, v, fail)) return def;
    v = lower(v);
    return v == "1" || v == "true" || v == "yes";
}

int CStream::printf(const char* fmt, ...)
{
    if (!m_out) return 0;
    va_list ap;
    va_start(ap, fmt);
    const int n = vfprintf(m_out, fmt, ap);
    va_end(ap);
    fflush(m_out);
    return n;
}
CFileGZInputStream::CFileGZInputStream(const std::string& file) { open(file); }
CFileGZInputStream::~CFileGZInputStream() { delete log; }
bool CFileGZInputStream::open(const std::string& file)
{
    m_name = file;
    if (!system::fileExists(file)) THROW_EXCEPTION("cannot open " + file);
    return true;
}
CFileOutputStream::CFileOutputStream(const std::string& file, bool append) { open(file, append); }
CFileOutputStream::~CFileOutputStream() { close(); }
bool CFileOutputStream::open(const std::string& file, bool append)
{
    close();
    m_name = file;
    m_out = fopen(file.c_str(), append ? "a" : "w");
    return m_out != nullptr;
}
void CFileOutputStream::close()
{
    if (m_out) fclose(m_out);
    m_out = nullptr;
}
}  // namespace utils

// ---- poses, math, random ------------------------------------------------------------------------------------------------
namespace poses {
CPose2D CPose2D::operator+(const CPose2D& b) const
{
    const double c = cos(m_phi), s = sin(m_phi);
    return CPose2D(m_x + b.m_x * c - b.m_y * s, m_y + b.m_x * s + b.m_y * c, wrap_pi(m_phi + b.m_phi));
}
CPose2D CPose2D::operator-(const CPose2D& b) const
{
    const double c = cos(b.m_phi), s = sin(b.m_phi), dx = m_x - b.m_x, dy = m_y - b.m_y;
    return CPose2D(dx * c + dy * s, -dx * s + dy * c, wrap_pi(m_phi - b.m_phi));
}
std::string CPose2D::asString() const { return mrpt::format("[%f %f %f]", m_x, m_y, utils::RAD2DEG(m_phi)); }
std::ostream& operator<<(std::ostream& o, const CPose2D& p)
{
    return o << mrpt::format("(%.03f,%.03f,%.02fdeg)", p.x(), p.y(), utils::RAD2DEG(p.phi()));
}
}  // namespace poses

namespace math {
void CMatrixDouble::loadFromTextFile(const std::string& file)
{
    std::ifstream in(file.c_str());
    if (!in) THROW_EXCEPTION("cannot open " + file);
    std::vector<std::vector<double> > rows;
    std::string line;
    while (std::getline(in, line)) {
        line = trim(line);
        if (line.empty() || line[0] == '%' || line[0] == '#') continue;
        for (size_t i = 0; i < line.size(); ++i) if (line[i] == ',' || line[i] == ';') line[i] = ' ';
        std::istringstream is(line);
        std::vector<double> r;
        double v;
        while (is >> v) r.push_back(v);
        if (!r.empty()) rows.push_back(r);
    }
    const size_t c = rows.empty() ? 0 : rows[0].size();
    setSize(rows.size(), c);
    for (size_t i = 0; i < rows.size(); ++i) {
        if (rows[i].size() != c) THROW_EXCEPTION("rows of different lengths in " + file);
        for (size_t j = 0; j < c; ++j) (*this)(i, j) = rows[i][j];
    }
}
double CMatrixDouble::det() const
{
    if (m_r != m_c) THROW_EXCEPTION("determinant of a non-square matrix");
    std::vector<double> a = m_v;
    const size_t n = m_r;
    double d = 1.0;
    for (size_t k = 0; k < n; ++k) {
        size_t piv = k;
        for (size_t i = k + 1; i < n; ++i) if (fabs(a[i * n + k]) > fabs(a[piv * n + k])) piv = i;
        if (a[piv * n + k] == 0.0) return 0.0;
        if (piv != k) { for (size_t j = 0; j < n; ++j) std::swap(a[k * n + j], a[piv * n + j]); d = -d; }
        d *= a[k * n + k];
        for (size_t i = k + 1; i < n; ++i) {
            const double f = a[i * n + k] / a[k * n + k];
            for (size_t j = k; j < n; ++j) a[i * n + j] -= f * a[k * n + j];
        }
    }
    return d;
}
double round_10power(double v, int power10)
{
    const double f = pow(10.0, -power10);
    return floor(v * f + 0.5) / f;
}
void confidenceIntervals(const std::vector<double>& data, double& mean, double& lo, double& hi, double conf, size_t)
{
    mean = lo = hi = 0;
    if (data.empty()) return;
    std::vector<double> s = data;
    std::sort(s.begin(), s.end());
    double sum = 0;
    for (size_t i = 0; i < s.size(); ++i) sum += s[i];
    mean = sum / (double)s.size();
    const double q = 0.5 * conf * (double)(s.size() - 1);
    lo = s[(size_t)floor(q)];
    hi = s[s.size() - 1 - (size_t)floor(q)];
}
}  // namespace math

namespace random {
CRandomGenerator randomGenerator;
static uint64_t splitmix(uint64_t& s)
{
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
void CRandomGenerator::randomize()
{
    const char* fixed = getenv("LSA_MRPT_SEED");          // a fixed seed makes a run repeatable (tests)
    randomize(fixed ? (uint32_t)strtoul(fixed, nullptr, 0) : (uint32_t)time(nullptr));
}
void CRandomGenerator::randomize(uint32_t seed) { m_state = 0x9E3779B97F4A7C15ull ^ ((uint64_t)seed << 17); m_have = false; }
double CRandomGenerator::drawUniform(double lo, double hi)
{
    const double u = (double)(splitmix(m_state) >> 11) * (1.0 / 9007199254740992.0);
    return lo + (hi - lo) * u;
}
double CRandomGenerator::drawGaussian1D_normalized()
{
    if (m_have) { m_have = false; return m_spare; }
    double u, v, s;
    do { u = drawUniform(-1, 1); v = drawUniform(-1, 1); s = u * u + v * v; } while (s >= 1.0 || s == 0.0);
    const double f = sqrt(-2.0 * log(s) / s);
    m_spare = v * f; m_have = true;
    return u * f;
}
}  // namespace random

// ---- opengl ---------------------------------------------------------------------------------------------------------------
namespace opengl {
CSetOfObjectsPtr lsa_make_set_of_objects() { return CSetOfObjects::Create(); }
void COpenGLScene::writeToFile(const std::string& file) const
{
    FILE* f = fopen(file.c_str(), "w");
    if (!f) return;
    fprintf(f, "# scene inventory (no MRPT serialisation here): kind name x y z yaw\n");
    for (size_t i = 0; i < m_objs.size(); ++i) {
        const CPose3D& p = m_objs[i]->getPose();
        fprintf(f, "%s \"%s\" %.6f %.6f %.6f %.6f\n", m_objs[i]->kind(), m_objs[i]->getName().c_str(), p.x, p.y, p.z, p.yaw);
    }
    fclose(f);
}
}  // namespace opengl

// ---- slam: observations, actions, the log ----------------------------------------------------------------------------------
namespace slam {

bool CSensoryFrame::insertObservationsInto(CSimplePointsMap* map) const
{
    if (!map) return false;
    for (size_t k = 0; k < m_obs.size(); ++k) {
        const CObservation2DRangeScan* s = dynamic_cast<const CObservation2DRangeScan*>(m_obs[k].pointer());
        if (!s) continue;
        const size_t n = s->scan.size();
        const double step = n > 1 ? (double)s->aperture / (double)(n - 1) : 0.0;
        for (size_t i = 0; i < n; ++i) {
            if (!s->validRange[i]) continue;
            const double a = (double)s->startAngle + step * (double)i;
            map->insertPoint((float)(s->scan[i] * cos(a)), (float)(s->scan[i] * sin(a)));
        }
    }
    return true;
}

void CActionRobotMovement2D::computeFromOdometry(const CPose2D& inc, const TMotionModelOptions& o)
{
    motionModelConfiguration = o;
    poseChange->mean = inc;
    const double d = inc.norm(), t = fabs(inc.phi());
    poseChange->std_xy = o.gausianModel.minStdXY + o.gausianModel.a1 * d + o.gausianModel.a2 * t;
    poseChange->std_phi = o.gausianModel.minStdPHI + o.gausianModel.a3 * d + o.gausianModel.a4 * t;
}

// CARMEN log lines (the formats CPolarMatch::readCarmon reads, src/zhpsm/ZHPolar_Match.cpp:172-284):
//   ROBOTLASER1 type start fov res max_range accuracy remission n r[n] m rem[m] lx ly lth ox oy oth ... timestamp host logger_ts
//   FLASER n r[n] lx ly lth ox oy oth timestamp host logger_ts                     (180 degrees, start -90)
bool lsa_parse_carmen_laser(const std::string& line, CObservation2DRangeScan& scan, double lp[3], double op[3])
{
    std::istringstream is(line);
    std::string tag;
    if (!(is >> tag)) return false;
    int n = 0;
    double start = -1.5707963267948966, fov = 3.14159265358979323846, max_range = 81.0;
    if (tag == "ROBOTLASER1") {
        int type, rem;
        double res, acc;
        if (!(is >> type >> start >> fov >> res >> max_range >> acc >> rem >> n)) return false;
    } else if (tag == "FLASER") {
        if (!(is >> n)) return false;
    } else {
        return false;
    }
    if (n <= 0 || n > 100000) return false;
    scan.scan.resize((size_t)n);
    scan.validRange.resize((size_t)n);
    for (int i = 0; i < n; ++i) {
        double r;
        if (!(is >> r)) return false;
        scan.scan[(size_t)i] = (float)r;
        scan.validRange[(size_t)i] = (r > 0.02 && r < max_range - 0.01) ? 1 : 0;
    }
    if (tag == "ROBOTLASER1") {
        int m = 0;
        if (!(is >> m)) return false;
        for (int i = 0; i < m; ++i) { double v; if (!(is >> v)) return false; }
    }
    if (!(is >> lp[0] >> lp[1] >> lp[2] >> op[0] >> op[1] >> op[2])) return false;
    double ts = 0;
    if (tag == "ROBOTLASER1") { double skip; for (int i = 0; i < 5; ++i) is >> skip; }
    is >> ts;
    scan.timestamp = (TTimeStamp)(ts * 1e7);
    scan.aperture = (float)fov;
    scan.startAngle = (float)start;
    scan.maxRange = (float)max_range;
    scan.rightToLeft = true;
    scan.sensorLabel = tag;
    return true;
}

bool CRawlog::getActionObservationPairOrObservation(CFileGZInputStream& in, CActionCollectionPtr& action,
                                                    CSensoryFramePtr& observations, CObservationPtr& observation,
                                                    size_t& rawlogEntry)
{
    action.clear(); observations.clear(); observation.clear();
    if (!in.log) {
        in.log = new std::ifstream(in.fileName().c_str());
        if (!*in.log) THROW_EXCEPTION("cannot open the log " + in.fileName());
    }
    std::string line;
    while (std::getline(*in.log, line)) {
        CObservation2DRangeScanPtr scan = CObservation2DRangeScan::Create();
        double lp[3], op[3];
        if (!lsa_parse_carmen_laser(line, *scan, lp, op)) continue;
        const CPose2D cur(op[0], op[1], op[2]);
        const CPose2D inc = in.have_prev ? cur - CPose2D(in.prev_odo[0], in.prev_odo[1], in.prev_odo[2]) : CPose2D();
        in.have_prev = true;
        in.prev_odo[0] = op[0]; in.prev_odo[1] = op[1]; in.prev_odo[2] = op[2];
        action = CActionCollection::Create();
        CActionRobotMovement2D mov;
        mov.computeFromOdometry(inc, CActionRobotMovement2D::TMotionModelOptions());
        action->insert(mov);
        observations = CSensoryFrame::Create();
        observations->insert(scan);
        rawlogEntry += 2;                                  // an action and a sensory frame, as in an MRPT rawlog
        return true;
    }
    return false;
}

void CSimpleMap::readFromFile(const std::string& file)
{
    std::ifstream in(file.c_str());
    if (!in) THROW_EXCEPTION("cannot open " + file);
    entries.clear();
    std::string line;
    while (std::getline(in, line)) {
        Entry e;
        e.scan = CObservation2DRangeScan::Create();
        double lp[3], op[3];
        if (!lsa_parse_carmen_laser(line, *e.scan, lp, op)) continue;
        e.pose = CPose2D(lp[0], lp[1], lp[2]);
        entries.push_back(e);
    }
}

// ---- maps -------------------------------------------------------------------------------------------------------------
void TSetOfMetricMapInitializers::loadFromConfigFile(const CConfigFileBase& cfg, const std::string& sec)
{
    grids.clear();
    const int n = cfg.read_int(sec, "occupancyGrid_count", 1);
    for (int i = 0; i < n; ++i) {
        const std::string sub = sec + mrpt::format("_occupancyGrid_%02u_creationOpts", (unsigned)i);
        TMetricMapInitializer g;
        g.extent_given = cfg.sectionExists(sub) && !cfg.read_string(sub, "min_x", "").empty();
        g.min_x = cfg.read_float(sub, "min_x", g.min_x); g.max_x = cfg.read_float(sub, "max_x", g.max_x);
        g.min_y = cfg.read_float(sub, "min_y", g.min_y); g.max_y = cfg.read_float(sub, "max_y", g.max_y);
        g.resolution = cfg.read_float(sub, "resolution", g.resolution);
        grids.push_back(g);
    }
}
void TSetOfMetricMapInitializers::dumpToConsole() const
{
    printf("metric maps: %u occupancy grid(s)\n", (unsigned)grids.size());
    for (size_t i = 0; i < grids.size(); ++i)
        printf("  grid %u: resolution %.3f m, extent %s [%.1f, %.1f] x [%.1f, %.1f]\n", (unsigned)i, grids[i].resolution,
               grids[i].extent_given ? "given" : "fitted to the data, default", grids[i].min_x, grids[i].max_x, grids[i].min_y, grids[i].max_y);
}

struct lsa_gridmap_impl {
    lsa_field* field = nullptr;
    float* d_prob = nullptr;
    ~lsa_gridmap_impl() { if (field) lsa_field_destroy(field); if (d_prob) lsa_free(d_prob); }
};

COccupancyGridMap2D::COccupancyGridMap2D(float a, float b, float c, float d, float r) { setSize(a, b, c, d, r); }
COccupancyGridMap2D::~COccupancyGridMap2D() { delete m_dev; }
void COccupancyGridMap2D::setSize(float min_x, float max_x, float min_y, float max_y, float res)
{
    m_res = res > 0 ? res : 0.1f;
    m_xmin = m_res * std::floor(min_x / m_res); m_ymin = m_res * std::floor(min_y / m_res);
    m_sx = (unsigned)std::ceil((max_x - m_xmin) / m_res); m_sy = (unsigned)std::ceil((max_y - m_ymin) / m_res);
    if (m_sx < 2) m_sx = 2;
    if (m_sy < 2) m_sy = 2;
    m_xmax = m_xmin + m_res * (float)m_sx; m_ymax = m_ymin + m_res * (float)m_sy;
    m_occ.assign((size_t)m_sx * m_sy, 0.5f);
    delete m_dev; m_dev = nullptr;
}
void COccupancyGridMap2D::computeEntropy(TEntropyInfo& info) const
{
    info = TEntropyInfo();
    for (size_t i = 0; i < m_occ.size(); ++i) {
        const double p = m_occ[i];
        if (fabs(p - 0.5) < 1e-3) continue;
        info.effectiveMappedCells++;
        if (p > 0 && p < 1) info.H += -p * log2(p) - (1 - p) * log2(1 - p);
    }
    info.effectiveMappedArea = (double)info.effectiveMappedCells * (double)m_res * (double)m_res;
    info.I = (double)info.effectiveMappedCells - info.H;
    if (info.effectiveMappedCells) { info.mean_H = info.H / info.effectiveMappedCells; info.mean_I = info.I / info.effectiveMappedCells; }
}
bool COccupancyGridMap2D::saveAsBitmapFile(const std::string& file) const
{
    FILE* f = fopen(file.c_str(), "wb");
    if (!f) return false;
    fprintf(f, "P5\n# limits %.9g %.9g %.9g %.9g %.9g\n%u %u\n255\n", (double)m_xmin, (double)m_xmax, (double)m_ymin, (double)m_ymax,
            (double)m_res, m_sx, m_sy);
    std::vector<unsigned char> row(m_sx);
    for (int y = (int)m_sy - 1; y >= 0; --y) {                         // image rows top-down = y descending
        for (unsigned x = 0; x < m_sx; ++x) row[x] = (unsigned char)std::lround(255.0 * (1.0 - m_occ[(size_t)x * m_sy + (size_t)y]));
        fwrite(row.data(), 1, row.size(), f);
    }
    fclose(f);
    return true;
}
void COccupancyGridMap2D::readFromFile(const std::string& file)
{
    std::ifstream in(file.c_str(), std::ios::binary);
    std::string magic, line;
    if (!in || !std::getline(in, magic) || trim(magic) != "P5") THROW_EXCEPTION("not a grid map of this build (binary PGM with a '# limits' comment): " + file);
    double xmin = 0, xmax = 0, ymin = 0, ymax = 0, res = 0;
    bool have = false;
    unsigned w = 0, h = 0, maxv = 0;
    while (std::getline(in, line)) {
        line = trim(line);
        if (line.empty()) continue;
        if (line[0] == '#') {
            if (sscanf(line.c_str(), "# limits %lf %lf %lf %lf %lf", &xmin, &xmax, &ymin, &ymax, &res) == 5) have = true;
            continue;
        }
        if (!w) { if (sscanf(line.c_str(), "%u %u", &w, &h) != 2) THROW_EXCEPTION("bad PGM header in " + file); continue; }
        maxv = (unsigned)strtoul(line.c_str(), nullptr, 10);
        break;
    }
    if (!have || !w || !h || maxv != 255) THROW_EXCEPTION("grid map without geometry comment: " + file);
    // the geometry as written (recomputing the cell counts from the limits could differ by one in float)
    m_res = (float)res; m_xmin = (float)xmin; m_ymin = (float)ymin; m_xmax = (float)xmax; m_ymax = (float)ymax;
    m_sx = w; m_sy = h;
    if (!(m_res > 0) || fabs((xmax - xmin) / res - (double)w) > 0.51 || fabs((ymax - ymin) / res - (double)h) > 0.51)
        THROW_EXCEPTION("grid map size does not match its limits: " + file);
    m_occ.assign((size_t)m_sx * m_sy, 0.5f);
    delete m_dev; m_dev = nullptr;
    std::vector<unsigned char> row(w);
    for (int y = (int)h - 1; y >= 0; --y) {
        in.read((char*)row.data(), (std::streamsize)w);
        if (!in) THROW_EXCEPTION("truncated grid map " + file);
        for (unsigned x = 0; x < w; ++x) m_occ[(size_t)x * m_sy + (size_t)y] = 1.0f - (float)row[x] / 255.0f;
    }
}
void COccupancyGridMap2D::getAs3DObject(opengl::CSetOfObjectsPtr& out) const
{
    if (!out) out = opengl::CSetOfObjects::Create();
    out->setName("gridmap");
}
void COccupancyGridMap2D::buildFromSimpleMap(const CSimpleMap& sm, bool fit_extent)
{
    if (sm.entries.empty()) return;
    // end points in the map frame
    std::vector<std::vector<float> > gx(sm.entries.size()), gy(sm.entries.size());
    float lo_x = 1e30f, hi_x = -1e30f, lo_y = 1e30f, hi_y = -1e30f;
    for (size_t k = 0; k < sm.entries.size(); ++k) {
        const CObservation2DRangeScan& s = *sm.entries[k].scan;
        const CPose2D& p = sm.entries[k].pose;
        const size_t n = s.scan.size();
        const double step = n > 1 ? (double)s.aperture / (double)(n - 1) : 0.0;
        for (size_t i = 0; i < n; ++i) {
            if (!s.validRange[i]) continue;
            const double a = p.phi() + (double)s.startAngle + step * (double)i;
            const float x = (float)(p.x() + s.scan[i] * cos(a)), y = (float)(p.y() + s.scan[i] * sin(a));
            gx[k].push_back(x); gy[k].push_back(y);
            lo_x = std::min(lo_x, x); hi_x = std::max(hi_x, x); lo_y = std::min(lo_y, y); hi_y = std::max(hi_y, y);
        }
        lo_x = std::min(lo_x, (float)p.x()); hi_x = std::max(hi_x, (float)p.x());
        lo_y = std::min(lo_y, (float)p.y()); hi_y = std::max(hi_y, (float)p.y());
    }
    if (fit_extent) setSize(lo_x - 1.0f, hi_x + 1.0f, lo_y - 1.0f, hi_y + 1.0f, m_res);
    lsa_ctx* c = device_ctx();
    lsa_grid* g = nullptr;
    lsa_ok(lsa_grid_create(&g, c, (int)m_sx, (int)m_sy, 0, 0, m_xmin, m_ymin, (double)m_res), "lsa_grid_create");
    for (size_t k = 0; k < sm.entries.size(); ++k) {
        if (gx[k].empty()) continue;
        lsa_ok(lsa_grid_update_host(g, gx[k].data(), gy[k].data(), (int)gx[k].size(), (float)sm.entries[k].pose.x(),
                                    (float)sm.entries[k].pose.y()), "lsa_grid_update_host");
    }
    lsa_ok(lsa_grid_compute_prob(g), "lsa_grid_compute_prob");
    std::vector<float> prob(m_occ.size()), sum(m_occ.size());
    lsa_ok(lsa_grid_download(g, 2, prob.data()), "lsa_grid_download");
    lsa_ok(lsa_grid_download(g, 1, sum.data()), "lsa_grid_download");
    lsa_grid_destroy(g);
    for (size_t i = 0; i < m_occ.size(); ++i) m_occ[i] = sum[i] > 0 ? prob[i] : 0.5f;   // never crossed by a ray: unknown
    delete m_dev; m_dev = nullptr;
}
lsa_gridmap_impl* COccupancyGridMap2D::device() const
{
    if (m_dev) return m_dev;
    lsa_ctx* c = device_ctx();
    m_dev = new lsa_gridmap_impl;
    lsa_ok(lsa_field_create(&m_dev->field, c, (int)m_sx, (int)m_sy, 0, 0, m_xmin, m_ymin, (double)m_res), "lsa_field_create");
    lsa_ok(lsa_ctx_malloc(c, (void**)&m_dev->d_prob, sizeof(float) * m_occ.size()), "lsa_malloc");
    lsa_ok(lsa_memcpy_h2d(c, m_dev->d_prob, m_occ.data(), sizeof(float) * m_occ.size()), "lsa_memcpy_h2d");
    lsa_ok(lsa_field_build(m_dev->field, m_dev->d_prob, 1.0f, 0.3f, 0.5f), "lsa_field_build");
    return m_dev;
}

void CMultiMetricMap::setListOfMaps(const TSetOfMetricMapInitializers* init)
{
    m_gridMaps.clear(); m_fit.clear();
    if (!init) return;
    for (size_t i = 0; i < init->grids.size(); ++i) {
        const TMetricMapInitializer& g = init->grids[i];
        m_gridMaps.push_back(COccupancyGridMap2DPtr(new COccupancyGridMap2D(g.min_x, g.max_x, g.min_y, g.max_y, g.resolution)));
        m_fit.push_back(!g.extent_given);
    }
}
void CMultiMetricMap::loadFromProbabilisticPosesAndObservations(const CSimpleMap& sm)
{
    for (size_t i = 0; i < m_gridMaps.size(); ++i) m_gridMaps[i]->buildFromSimpleMap(sm, m_fit[i]);
}

// ---- particle filter ------------------------------------------------------------------------------------------------------
void TKLDParams::loadFromConfigFile(const CConfigFileBase& cfg, const std::string& sec)
{
    KLD_binSize_XY = cfg.read_double(sec, "KLD_binSize_XY", KLD_binSize_XY);
    KLD_binSize_PHI = utils::DEG2RAD(cfg.read_double(sec, "KLD_binSize_PHI_deg", utils::RAD2DEG(KLD_binSize_PHI)));
    KLD_delta = cfg.read_double(sec, "KLD_delta", KLD_delta);
    KLD_epsilon = cfg.read_double(sec, "KLD_epsilon", KLD_epsilon);
    KLD_minSampleSize = (unsigned)cfg.read_int(sec, "KLD_minSampleSize", (int)KLD_minSampleSize);
    KLD_maxSampleSize = (unsigned)cfg.read_int(sec, "KLD_maxSampleSize", (int)KLD_maxSampleSize);
}
void CParticleFilter::TParticleFilterOptions::loadFromConfigFile(const CConfigFileBase& cfg, const std::string& sec)
{
    adaptiveSampleSize = cfg.read_bool(sec, "adaptiveSampleSize", adaptiveSampleSize);
    BETA = cfg.read_double(sec, "BETA", BETA);
    sampleSize = (unsigned)cfg.read_int(sec, "sampleSize", (int)sampleSize);
    powFactor = cfg.read_double(sec, "powFactor", powFactor);
    PF_algorithm = cfg.read_int(sec, "PF_algorithm", PF_algorithm);
    resamplingMethod = cfg.read_int(sec, "resamplingMethod", resamplingMethod);
    max_loglikelihood_dyn_range = cfg.read_double(sec, "max_loglikelihood_dyn_range", max_loglikelihood_dyn_range);
}
void CParticleFilter::TParticleFilterOptions::dumpToConsole() const
{
    printf("particle filter: standard proposal, systematic resampling when ESS < %.2f (file: PF_algorithm %d, resamplingMethod %d, "
           "adaptiveSampleSize %d - the sample size is fixed here), powFactor %.2f\n", BETA, PF_algorithm, resamplingMethod,
           (int)adaptiveSampleSize, powFactor);
}

struct lsa_mcl_impl {
    size_t n = 0;
    float *d_poses = nullptr, *d_tmp = nullptr, *d_normals = nullptr, *d_score = nullptr, *d_logw = nullptr, *d_w = nullptr;
    float *d_sx = nullptr, *d_sy = nullptr;
    int32_t* d_parent = nullptr;
    size_t scan_cap = 0;
    std::vector<float> h_poses, h_logw, h_score, h_normals;
    std::mt19937_64 rng;
    void release()
    {
        float** ps[] = {&d_poses, &d_tmp, &d_normals, &d_score, &d_logw, &d_w, &d_sx, &d_sy};
        for (float** p : ps) { if (*p) lsa_free(*p); *p = nullptr; }
        if (d_parent) lsa_free(d_parent);
        d_parent = nullptr; scan_cap = 0;
    }
    void resize(size_t m)
    {
        if (m == n && d_poses) return;
        release();
        n = m;
        const size_t b3 = sizeof(float) * 3 * n, b1 = sizeof(float) * n;
        lsa_ctx* c = device_ctx();          // (the buffers have to live on the device the context launches on, whatever the thread's current one is)
        lsa_ok(lsa_ctx_malloc(c, (void**)&d_poses, b3), "lsa_malloc"); lsa_ok(lsa_ctx_malloc(c, (void**)&d_tmp, b3), "lsa_malloc");
        lsa_ok(lsa_ctx_malloc(c, (void**)&d_normals, b3), "lsa_malloc"); lsa_ok(lsa_ctx_malloc(c, (void**)&d_score, b1), "lsa_malloc");
        lsa_ok(lsa_ctx_malloc(c, (void**)&d_logw, b1), "lsa_malloc"); lsa_ok(lsa_ctx_malloc(c, (void**)&d_w, b1), "lsa_malloc");
        lsa_ok(lsa_ctx_malloc(c, (void**)&d_parent, sizeof(int32_t) * n), "lsa_malloc");
        h_poses.assign(3 * n, 0.0f); h_logw.assign(n, 0.0f); h_score.assign(n, 0.0f); h_normals.assign(3 * n, 0.0f);
    }
    void upload_poses()
    {
        lsa_ok(lsa_memcpy_h2d(device_ctx(), d_poses, h_poses.data(), sizeof(float) * 3 * n), "lsa_memcpy_h2d");
    }
    void download_poses()
    {
        lsa_ok(lsa_memcpy_d2h(device_ctx(), h_poses.data(), d_poses, sizeof(float) * 3 * n), "lsa_memcpy_d2h");
    }
    // normalised linear weights of the host mirror
    void weights(std::vector<double>& w) const
    {
        w.resize(n);
        double mx = -1e300, s = 0;
        for (size_t i = 0; i < n; ++i) mx = std::max(mx, (double)h_logw[i]);
        for (size_t i = 0; i < n; ++i) { w[i] = exp((double)h_logw[i] - mx); s += w[i]; }
        for (size_t i = 0; i < n; ++i) w[i] /= s;
    }
};

CMonteCarloLocalization2D::CMonteCarloLocalization2D(size_t M) : m_impl(new lsa_mcl_impl)
{
    const char* fixed = getenv("LSA_MRPT_SEED");
    m_impl->rng.seed(fixed ? strtoull(fixed, nullptr, 0) : (unsigned long long)time(nullptr));
    m_impl->resize(M ? M : 1);
    m_impl->upload_poses();
}
CMonteCarloLocalization2D::~CMonteCarloLocalization2D()
{
    m_impl->release();
    delete m_impl;
}
size_t CMonteCarloLocalization2D::size() const { return m_impl->n; }
void CMonteCarloLocalization2D::resetUniform(double x0, double x1, double y0, double y1, double p0, double p1, int count)
{
    lsa_mcl_impl& I = *m_impl;
    if (count > 0) I.resize((size_t)count);
    std::uniform_real_distribution<double> u(0.0, 1.0);
    for (size_t i = 0; i < I.n; ++i) {
        I.h_poses[3 * i] = (float)(x0 + (x1 - x0) * u(I.rng));
        I.h_poses[3 * i + 1] = (float)(y0 + (y1 - y0) * u(I.rng));
        I.h_poses[3 * i + 2] = (float)(p0 + (p1 - p0) * u(I.rng));
        I.h_logw[i] = 0.0f;
    }
    I.upload_poses();
}
void CMonteCarloLocalization2D::resetUniformFreeSpace(COccupancyGridMap2D* map, double thr, int count, double x0, double x1,
                                                      double y0, double y1, double p0, double p1)
{
    ASSERT_(map != nullptr)
    lsa_mcl_impl& I = *m_impl;
    if (count > 0) I.resize((size_t)count);
    std::vector<std::pair<int, int> > cells;
    for (unsigned cx = 0; cx < map->getSizeX(); ++cx)
        for (unsigned cy = 0; cy < map->getSizeY(); ++cy) {
            const double x = map->getXMin() + (cx + 0.5) * map->getResolution(), y = map->getYMin() + (cy + 0.5) * map->getResolution();
            if (x < x0 || x > x1 || y < y0 || y > y1) continue;
            if (map->getCell((int)cx, (int)cy) >= thr) cells.push_back(std::make_pair((int)cx, (int)cy));
        }
    if (cells.empty()) THROW_EXCEPTION("resetUniformFreeSpace: no free cell of the map inside the given area");
    std::uniform_real_distribution<double> u(0.0, 1.0);
    for (size_t i = 0; i < I.n; ++i) {
        const std::pair<int, int>& c = cells[(size_t)(u(I.rng) * (double)cells.size()) % cells.size()];
        I.h_poses[3 * i] = (float)(map->getXMin() + (c.first + u(I.rng)) * map->getResolution());
        I.h_poses[3 * i + 1] = (float)(map->getYMin() + (c.second + u(I.rng)) * map->getResolution());
        I.h_poses[3 * i + 2] = (float)(p0 + (p1 - p0) * u(I.rng));
        I.h_logw[i] = 0.0f;
    }
    I.upload_poses();
}
CPose2D CMonteCarloLocalization2D::getParticlePose(size_t i) const
{
    return CPose2D(m_impl->h_poses[3 * i], m_impl->h_poses[3 * i + 1], m_impl->h_poses[3 * i + 2]);
}
double CMonteCarloLocalization2D::getW(size_t i) const { return m_impl->h_logw[i]; }
void CMonteCarloLocalization2D::getMean(CPose2D& mean) const
{
    math::CMatrixDouble33 cov;
    getCovarianceAndMean(cov, mean);
}
void CMonteCarloLocalization2D::getCovarianceAndMean(math::CMatrixDouble33& cov, CPose2D& mean) const
{
    const lsa_mcl_impl& I = *m_impl;
    std::vector<double> w;
    I.weights(w);
    double mx = 0, my = 0, sc = 0, ss = 0;
    for (size_t i = 0; i < I.n; ++i) {
        mx += w[i] * I.h_poses[3 * i]; my += w[i] * I.h_poses[3 * i + 1];
        sc += w[i] * cos((double)I.h_poses[3 * i + 2]); ss += w[i] * sin((double)I.h_poses[3 * i + 2]);
    }
    const double mp = atan2(ss, sc);
    mean = CPose2D(mx, my, mp);
    double c[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (size_t i = 0; i < I.n; ++i) {
        const double d[3] = {I.h_poses[3 * i] - mx, I.h_poses[3 * i + 1] - my, wrap_pi((double)I.h_poses[3 * i + 2] - mp)};
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) c[a][b] += w[i] * d[a] * d[b];
    }
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) cov(a, b) = c[a][b];
}
void CMonteCarloLocalization2D::getCovariance(math::CMatrixDouble& cov) const
{
    math::CMatrixDouble33 c;
    CPose2D m;
    getCovarianceAndMean(c, m);
    cov.setSize(3, 3);
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) cov(a, b) = c(a, b);
}
double CMonteCarloLocalization2D::ESS() const
{
    std::vector<double> w;
    m_impl->weights(w);
    double s2 = 0;
    for (size_t i = 0; i < w.size(); ++i) s2 += w[i] * w[i];
    return 1.0 / (s2 * (double)w.size());
}
bool CMonteCarloLocalization2D::saveToTextFile(const std::string& file) const
{
    FILE* f = fopen(file.c_str(), "w");
    if (!f) return false;
    for (size_t i = 0; i < m_impl->n; ++i)
        fprintf(f, "%f %f %f %e\n", (double)m_impl->h_poses[3 * i], (double)m_impl->h_poses[3 * i + 1], (double)m_impl->h_poses[3 * i + 2],
                (double)m_impl->h_logw[i]);
    fclose(f);
    return true;
}

void CParticleFilter::executeOn(CMonteCarloLocalization2D& pdf, const CActionCollection* action, const CSensoryFrame* sf,
                                TParticleFilterStats* stats)
{
    lsa_mcl_impl& I = *pdf.impl();
    lsa_ctx* c = device_ctx();
    const size_t n = I.n;
    // ---- prediction: every particle is composed with its own draw of the odometry increment
    CActionRobotMovement2DPtr mov = action ? action->getBestMovementEstimation() : CActionRobotMovement2DPtr();
    if (mov) {
        std::normal_distribution<float> g(0.0f, 1.0f);
        for (size_t i = 0; i < 3 * n; ++i) I.h_normals[i] = g(I.rng);
        lsa_ok(lsa_memcpy_h2d(c, I.d_normals, I.h_normals.data(), sizeof(float) * 3 * n), "lsa_memcpy_h2d");
        const CPose2D m = mov->poseChange->getMeanVal();
        const double delta[3] = {m.x(), m.y(), m.phi()};
        const double sigma[3] = {mov->poseChange->std_xy, mov->poseChange->std_xy, mov->poseChange->std_phi};
        lsa_ok(lsa_pf_motion(c, (int)n, delta, sigma, I.d_normals, I.d_poses), "lsa_pf_motion");
    }
    // ---- update: log-likelihood of the first range scan of the sensory frame in the first grid map's likelihood field
    const CObservation2DRangeScan* scan = nullptr;
    for (size_t k = 0; sf && k < sf->size() && !scan; ++k) scan = dynamic_cast<const CObservation2DRangeScan*>(sf->getObservationByIndex(k).pointer());
    COccupancyGridMap2D* map = (pdf.options.metricMap && !pdf.options.metricMap->m_gridMaps.empty())
                                   ? pdf.options.metricMap->m_gridMaps[0].pointer() : nullptr;
    if (scan && map) {
        std::vector<float> sx, sy;
        const size_t m = scan->scan.size();
        const double step = m > 1 ? (double)scan->aperture / (double)(m - 1) : 0.0;
        for (size_t i = 0; i < m; ++i) {
            if (!scan->validRange[i]) continue;
            const double a = (double)scan->startAngle + step * (double)i;
            sx.push_back((float)(scan->scan[i] * cos(a))); sy.push_back((float)(scan->scan[i] * sin(a)));
        }
        if (!sx.empty()) {
            if (sx.size() > I.scan_cap) {
                if (I.d_sx) lsa_free(I.d_sx);
                if (I.d_sy) lsa_free(I.d_sy);
                I.scan_cap = sx.size() + 64;
                lsa_ok(lsa_ctx_malloc(c, (void**)&I.d_sx, sizeof(float) * I.scan_cap), "lsa_malloc");
                lsa_ok(lsa_ctx_malloc(c, (void**)&I.d_sy, sizeof(float) * I.scan_cap), "lsa_malloc");
            }
            lsa_ok(lsa_memcpy_h2d(c, I.d_sx, sx.data(), sizeof(float) * sx.size()), "lsa_memcpy_h2d");
            lsa_ok(lsa_memcpy_h2d(c, I.d_sy, sy.data(), sizeof(float) * sy.size()), "lsa_memcpy_h2d");
            const int decim = sx.size() > 120 ? (int)(sx.size() / 60) : 1;          // about sixty rays per scan
            lsa_ok(lsa_score_poses_std(map->device()->field, I.d_sx, I.d_sy, (int)sx.size(), decim, I.d_poses, (int)n, I.d_score),
                   "lsa_score_poses_std");
            lsa_ok(lsa_memcpy_d2h(c, I.h_score.data(), I.d_score, sizeof(float) * n), "lsa_memcpy_d2h");
            for (size_t i = 0; i < n; ++i) I.h_logw[i] += (float)(m_options.powFactor * (double)I.h_score[i]);
        }
    }
    // ---- effective sample size; systematic resampling below BETA
    lsa_ok(lsa_memcpy_h2d(c, I.d_logw, I.h_logw.data(), sizeof(float) * n), "lsa_memcpy_h2d");
    float mx = 0;
    int32_t arg = 0;
    lsa_ok(lsa_pose_argmax(c, I.d_logw, (int)n, &mx, &arg), "lsa_pose_argmax");
    double sum = 0, sumsq = 0;
    lsa_ok(lsa_pf_weights(c, I.d_logw, (int)n, mx, I.d_w, &sum, &sumsq), "lsa_pf_weights");
    const double ess = sumsq > 0 ? sum * sum / sumsq / (double)n : 0.0;
    if (stats) {
        stats->ESS_beforeResample = ess;
        const double mean_w = 1.0 / (double)n;
        stats->weightsVariance_beforeResample = sum > 0 ? sumsq / (sum * sum) / (double)n - mean_w * mean_w : 0.0;
    }
    if (ess < m_options.BETA) {
        std::uniform_real_distribution<double> u(0.0, 1.0);
        lsa_ok(lsa_pf_resample(c, I.d_w, (int)n, u(I.rng), (int)n, 0, (int)n, I.d_parent), "lsa_pf_resample");
        lsa_ok(lsa_gather_poses(c, I.d_poses, I.d_parent, (int)n, I.d_tmp), "lsa_gather_poses");
        std::swap(I.d_poses, I.d_tmp);
        std::fill(I.h_logw.begin(), I.h_logw.end(), 0.0f);
    }
    I.download_poses();
}

}  // namespace slam
}  // namespace mrpt
