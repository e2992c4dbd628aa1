// CARMEN log reader of the C ABI (include/lsa_core.h: lsa_carmen_read).
// Replaces CPolarMatch::readCarmon / readFlirt, src/zhpsm/ZHPolar_Match.cpp:172-289:
// same record layout, the same strtok/atof token semantics, the same m->cm scaling,
// x,y from the raw range, min-range clamp, n == points or points-1 acceptance, and
// the same side effects on the parameter block (fi_min, fi_max, max_range taken
// from the first record).  Host-only text parsing; no device work.
#include "../../include/lsa_core.h"
#include "lsa_math.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

const double kD2R = 3.141592654 / 180.0;    // PM_D2R with the reference's M_PI (PolarParameter.h:38-39)

struct Tok {
    char* save = nullptr;
    char* next(char* first = nullptr) { return strtok_r(first, " ", &save); }
};

template <class T> T* dup_out(const std::vector<T>& v)
{
    T* p = (T*)malloc(sizeof(T) * (v.empty() ? 1 : v.size()));
    if (p && !v.empty()) memcpy(p, v.data(), sizeof(T) * v.size());
    return p;
}

} // namespace

extern "C" int lsa_carmen_read(lsa_laser_param* p, const char* filename, const char* tag,
                               int* n_scans, float** h_r, float** h_x, float** h_y,
                               int32_t** h_bad, float** h_pose)
{
    if (!p || !filename || !tag || !n_scans || !h_r || !h_x || !h_y || !h_bad || !h_pose) return LSA_E_ARG;
    FILE* f = fopen(filename, "r");
    if (!f) return LSA_E_IO;
    const int n = p->points;
    const bool poseless = strncmp(tag, "LMS", 3) == 0;       // in-house format, :204-211
    std::vector<float> R, X, Y, POSE, co(n), si(n), fi(n);
    std::vector<int32_t> BAD;
    std::vector<float> r(n, 0.f), x(n, 0.f), y(n, 0.f);      // the reader's reused scan (:188)
    std::vector<int32_t> bad(n, 0);
    float pose[3] = {0, 0, 0};
    lsa_param_tables(p, fi.data(), si.data(), co.data());

    std::vector<char> line(8192);
    bool first = true, failed = false;
    int count = 0;
    while (fgets(line.data(), (int)line.size(), f)) {
        size_t len = strlen(line.data());
        if (len && line[len - 1] == '\n') line[--len] = 0;
        else if (len == line.size() - 1) break;              // longer than the reference's 8192-byte buffer
        Tok t;
        char* w = t.next(line.data());
        if (!w || strcmp(w, tag) != 0) continue;             // :214-216
        t.next();                                            // laser type
        const double start = atof(t.next());
        const double fov = atof(t.next());
        t.next();                                            // resolution (parsed, unused: :231)
        const double max_range_m = atof(t.next());
        t.next();                                            // accuracy
        t.next();                                            // remission mode
        const int np = atoi(t.next());
        if (np == 0) continue;
        if (first) {                                         // :227-235
            first = false;
            p->fi_min = (float)start;
            p->fi_max = (float)(start + fov * kD2R);
            lsa_param_tables(p, fi.data(), si.data(), co.data());
            if (max_range_m * 100 != p->max_range) p->max_range = (int)(max_range_m * 100);
        }
        if (np != n - 1 && np != n) { failed = true; break; }    // :238-248
        for (int i = 0; i < np; ++i) {                       // :250-259
            r[i] = (float)(atof(t.next()) * 100.0);
            x[i] = r[i] * co[i];
            y[i] = r[i] * si[i];
            bad[i] = 0;
            if (r[i] < 10.0f) r[i] = (float)(p->max_range + 1);
        }
        if (!poseless) {                                     // :262-270
            t.next();                                        // remission count, skipped as one token
            pose[0] = (float)atof(t.next());
            pose[1] = (float)atof(t.next());
            pose[2] = (float)atof(t.next());
        }
        if (np == n - 1) { r[np] = (float)(p->max_range + 1); bad[np] = 0; }   // :276-279
        R.insert(R.end(), r.begin(), r.end());
        X.insert(X.end(), x.begin(), x.end());
        Y.insert(Y.end(), y.begin(), y.end());
        BAD.insert(BAD.end(), bad.begin(), bad.end());
        POSE.insert(POSE.end(), pose, pose + 3);
        ++count;
    }
    fclose(f);
    if (failed || count == 0) return LSA_E_IO;
    *n_scans = count;
    *h_r = dup_out(R); *h_x = dup_out(X); *h_y = dup_out(Y); *h_bad = dup_out(BAD); *h_pose = dup_out(POSE);
    return LSA_OK;
}

extern "C" void lsa_host_free(void* p) { free(p); }
