// Exercises the C++ drop-in classes (laser_slam_amd/host) the way the reference's own
// drivers use the originals (src/zhpsm/test.cpp, src/mapGraph/FlirterNode.cpp:255-375,
// src/drawmap/PMap2D.cpp:33-39) and prints results as hex floats for bit-exact
// comparison with the golden vectors (tests/test_gpu_host_dropin.py).
//   host_dropin_test match <laser> <log> <tag> <n_pairs>
//   host_dropin_test pmap <in.bin> <out.bin>
#include "zhpsm/ZHPolar_Match.h"
#include "zhpsm/point.h"
#include "zhcsm/ZHCanonical_Matcher.h"
#include "drawmap/PMap.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

static int run_match(const char* laser, const char* log, const char* tag, int n_pairs)
{
    CPolarMatch pm(laser);
    if (!pm.m_bReady) return 2;
    if (!pm.readCarmon(log, tag)) return 3;
    printf("param %d %d %a %a\n", pm.m_pParam->pm_l_points, pm.m_pParam->pm_max_range, pm.m_pParam->pm_fi_min, pm.m_pParam->pm_dfi);
    const int S = (int)pm.m_SickScans.size();
    if (n_pairs > S - 1) n_pairs = S - 1;

    // 1. FMatch chain exactly as CFliterNode::matchNodeFrontend uses it: each node keeps its
    //    own deep copy; the scan is preprocessed once when it is the current scan.
    {
        PMScan* prev = new PMScan(*pm.m_SickScans[0]);
        for (int b = 0; b < n_pairs; ++b) {
            PMScan* cur = new PMScan(*pm.m_SickScans[b + 1]);
            std::pair<float, float> e = pm.FMatch(prev, cur, b == 0);
            printf("fmatch %d %a %a %a %a %a\n", b, cur->rx, cur->ry, cur->th, e.first, e.second);
            delete prev;
            prev = cur;
        }
        delete prev;
    }
    // 2. explicit preprocess + pm_psm / pm_icp with catch(int), + pm_error_index2
    std::vector<PMScan*> pre;
    for (int k = 0; k <= n_pairs; ++k) {
        PMScan* s = new PMScan(*pm.m_SickScans[k]);
        pm.pm_preprocessScan(s);
        pre.push_back(s);
    }
    for (int which = 0; which < 2; ++which)
        for (int b = 0; b < n_pairs; ++b) {
            PMScan ref(*pre[b]), cur(*pre[b + 1]);
            ref.rx = ref.ry = ref.th = 0; cur.rx = cur.ry = cur.th = 0;
            int status = 0;
            float ret = 0;
            try { ret = which == 0 ? pm.pm_psm(&ref, &cur) : pm.pm_icp(&ref, &cur); }
            catch (int err) { status = err; }
            printf("%s %d %d %a %a %a %a", which == 0 ? "psm" : "icp", b, status, cur.rx, cur.ry, cur.th, ret);
            if (status == 0) {
                int na = 0;
                std::pair<float, float> e = pm.pm_error_index2(&ref, &cur, &na);
                printf(" %d %a %a", na, e.first, e.second);
            }
            printf("\n");
        }
    // 3. stage methods on pair 0 at a fixed pose
    {
        PMScan cur(*pre[1]);
        cur.rx = 3.5f; cur.ry = -2.25f; cur.th = 0.03f;
        const int n = pm.m_pParam->pm_l_points;
        std::vector<float> new_r(n);
        std::vector<int> new_bad(n);
        pm.pm_scan_project(&cur, new_r.data(), new_bad.data());
        unsigned h = 2166136261u;
        for (int i = 0; i < n; ++i) { unsigned u; memcpy(&u, &new_r[i], 4); h = (h ^ u) * 16777619u; h = (h ^ (unsigned)new_bad[i]) * 16777619u; }
        float dth = 0, dx = 0, dy = 0, err = 0;
        int so = 0, stt = 0;
        try { dth = pm.pm_orientation_search(pre[0], new_r.data(), new_bad.data()); } catch (int e) { so = e; }
        try { err = pm.pm_translation_estimation(pre[0], new_r.data(), new_bad.data(), PM_WEIGHTING_FACTOR, &dx, &dy); } catch (int e) { stt = e; }
        printf("stage %08x %d %a %d %a %a %a\n", h, so, dth, stt, dx, dy, err);
        int corridor = -1;
        try { corridor = pm.pm_is_corridor(pre[1]) ? 1 : 0; } catch (int e) { corridor = -e; }
        printf("corridor %d\n", corridor);
    }
    // 4. batched addition
    {
        std::vector<float> poses;
        std::vector<int> status;
        pm.matchConsecutive(1, poses, status);
        for (int b = 0; b < n_pairs; ++b) printf("batch %d %d %a %a %a\n", b, status[b], poses[3 * b], poses[3 * b + 1], poses[3 * b + 2]);
    }
    // 5. CCanonicalMatcher surface (parity unpinned: only behaviour is checked)
    {
        CCanonicalMatcher csm(pm.m_pParam);
        PMScan a(*pm.m_SickScans[0]), b(*pm.m_SickScans[1]);
        float r0 = csm.FMatchKeyFrame(&a);
        float r1 = csm.FMatchKeyFrame(&b);
        printf("csm %g %g %a %a %a\n", r0, r1, b.rx, b.ry, b.th);
    }
    // 6. pose algebra
    {
        zhpsm::OrientedPoint2D p(1.0, 2.0, 0.5), q(0.25, -0.5, 0.125);
        zhpsm::OrientedPoint2D s = p.oplus(q), d = p.ominus(s);
        printf("se2 %a %a %a %a %a %a\n", s.x, s.y, s.theta, d.x, d.y, d.theta);
    }
    for (size_t k = 0; k < pre.size(); ++k) delete pre[k];
    return 0;
}

// in.bin: int32 geom[4] (sx, sy, cx, cy), float off[2], double res, int32 S, int32 nb,
//         float gx[S*nb], gy[S*nb], robot[S*3] (x, y, n_valid)
// out.bin: float hit[sx*sy], sum[sx*sy], prob[sx*sy], int32 ncloud, float cx[], cy[]
static int run_pmap(const char* in, const char* out)
{
    FILE* f = fopen(in, "rb");
    if (!f) return 2;
    int geom[4], S, nb;
    float off[2];
    double res;
    if (fread(geom, 4, 4, f) != 4 || fread(off, 4, 2, f) != 2 || fread(&res, 8, 1, f) != 1 ||
        fread(&S, 4, 1, f) != 1 || fread(&nb, 4, 1, f) != 1) return 3;
    std::vector<float> gx((size_t)S * nb), gy((size_t)S * nb), rob((size_t)S * 3);
    if (fread(gx.data(), 4, gx.size(), f) != gx.size() || fread(gy.data(), 4, gy.size(), f) != gy.size() ||
        fread(rob.data(), 4, rob.size(), f) != rob.size()) return 3;
    fclose(f);
    CPMap map(0.95f);
    map.initializeMap(geom[0], geom[1], geom[2], geom[3], off[0], off[1], res);
    const auto t0 = std::chrono::steady_clock::now();
    for (int s = 0; s < S; ++s)          // the reference's per-scan signature, incl. the "finish" call
        map.updateMap(&gx[(size_t)s * nb], &gy[(size_t)s * nb], (int)rob[3 * s + 2], rob[3 * s], rob[3 * s + 1], 0.0f);
    map.flushPending();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    fprintf(stderr, "CPMap::updateMap x %d (host buffers in, PCIe inclusive): %.0f scans/s\n", S, S / dt);
    map.updateMap(0, 0, -1, 0, 0, 0);
    std::vector<float> cx, cy;
    map.getPointCloud(cx, cy);
    int ix, iy;
    map.mapPosfromPoint(rob[0], rob[1], ix, iy);
    FILE* g = fopen(out, "wb");
    for (int layer = 0; layer < 3; ++layer) {
        std::vector<std::vector<float> >& L = layer == 0 ? map.m_maphit : layer == 1 ? map.m_mapsum : map.m_mapprob;
        for (int x = 0; x < geom[0]; ++x) fwrite(L[x].data(), 4, geom[1], g);
    }
    int nc = (int)cx.size();
    fwrite(&nc, 4, 1, g);
    fwrite(cx.data(), 4, nc, g); fwrite(cy.data(), 4, nc, g);
    fwrite(&ix, 4, 1, g); fwrite(&iy, 4, 1, g);
    fclose(g);
    return 0;
}

// readers rawseed|sick|511 <laser> <file> <out.bin>: out = int32 count, then per record float r[n], x[n], y[n], int32 bad[n]
// (511: every line gives a record preceded by an int32 "parsed" flag; refused lines carry zeros)
static int run_readers(const char* kind, const char* laser, const char* file, const char* out)
{
    CPolarMatch m(laser);
    const int n = m.m_pParam->pm_l_points;
    FILE* g = fopen(out, "wb");
    if (!g) return 2;
    if (!strcmp(kind, "rawseed") || !strcmp(kind, "sick")) {
        if (!(kind[0] == 'r' ? m.readRawSeed(file) : m.readSICK(file))) return 3;
        int S = (int)m.m_SickScans.size();
        fwrite(&S, 4, 1, g);
        for (int k = 0; k < S; ++k) {
            PMScan* s = m.m_SickScans[k];
            fwrite(s->r, 4, n, g); fwrite(s->x, 4, n, g); fwrite(s->y, 4, n, g); fwrite(s->bad, 4, n, g);
        }
    } else {
        FILE* f = fopen(file, "r");
        if (!f) return 3;
        std::vector<char> line(1 << 16);
        std::vector<std::vector<char> > recs;
        int S = 0;
        long pos = ftell(g);
        fwrite(&S, 4, 1, g);
        while (fgets(line.data(), (int)line.size(), f)) {
            size_t L = strlen(line.data());
            while (L && (line[L - 1] == '\n' || line[L - 1] == '\r')) line[--L] = 0;
            if (!L) continue;
            PMScan scan(n);
            for (int i = 0; i < n; ++i) { scan.r[i] = scan.x[i] = scan.y[i] = 0; scan.bad[i] = 0; }
            int ok = m.construct511Data(line.data(), scan) ? 1 : 0;
            if (!ok) for (int i = 0; i < n; ++i) { scan.r[i] = scan.x[i] = scan.y[i] = 0; scan.bad[i] = 0; }
            fwrite(&ok, 4, 1, g);
            fwrite(scan.r, 4, n, g); fwrite(scan.x, 4, n, g); fwrite(scan.y, 4, n, g); fwrite(scan.bad, 4, n, g);
            ++S;
        }
        fclose(f);
        fseek(g, pos, SEEK_SET); fwrite(&S, 4, 1, g);
    }
    fclose(g);
    return 0;
}

// corridor <laser> <fi_min> <fi_max> <max_range> <in.bin> : in = int32 S, then per scan float r[n], int32 bad[n], int32 seg[n];
// prints one number per scan: 0 / 1, or 100 + the int pm_is_corridor throws.  Needs no device.
static int run_corridor(const char* laser, const char* fi_min, const char* fi_max, const char* max_range, const char* in)
{
    CPolarMatch m(laser);
    if (strcmp(fi_min, "-")) {
        m.m_pParam->pm_fi_min = (float)atof(fi_min); m.m_pParam->pm_fi_max = (float)atof(fi_max); m.m_pParam->pm_max_range = atoi(max_range);
        m.pm_init();
    }
    const int n = m.m_pParam->pm_l_points;
    FILE* f = fopen(in, "rb");
    if (!f) return 2;
    int S = 0;
    if (fread(&S, 4, 1, f) != 1) return 3;
    PMScan scan(n);
    for (int k = 0; k < S; ++k) {
        if (fread(scan.r, 4, n, f) != (size_t)n || fread(scan.bad, 4, n, f) != (size_t)n || fread(scan.seg, 4, n, f) != (size_t)n) return 3;
        int res;
        try { res = m.pm_is_corridor(&scan) ? 1 : 0; } catch (int e) { res = 100 + e; }
        printf("%d\n", res);
    }
    fclose(f);
    return 0;
}

// csmx <laser> <log> <tag>: CCanonicalMatcher's laser_data-typed overloads and FMatchKeyFrame2 with sm_icp's covariance
static int run_csmx(const char* laser, const char* log, const char* tag)
{
    CPolarMatch pm(laser);
    if (!pm.m_bReady || !pm.readCarmon(log, tag)) return 2;
    CCanonicalMatcher csm(pm.m_pParam);
    // 1. LDP overload == PMScan overload (the PMScan one is median filter + conversion + the LDP one)
    PMScan a(*pm.m_SickScans[0]), b(*pm.m_SickScans[1]);
    float r_pm = csm.FMatch(&a, &b, true);
    printf("pm %g %a %a %a\n", r_pm, b.rx, b.ry, b.th);
    LDP la = 0, lb = 0;
    csm.fromPMScan2LDP(&a, &la, csm.m_pParam);        // a, b are median filtered by now
    csm.fromPMScan2LDP(&b, &lb, csm.m_pParam);
    float r_ld = csm.FMatch(la, lb);
    printf("ld %g %a %a %a %d %d\n", r_ld, lb->true_pose[0], lb->true_pose[1], lb->true_pose[2], la->nrays, csm.output->valid);
    ld_free(la); ld_free(lb);
    // 2. key frames with the covariance
    csm.setCov(true);
    struct Mat { double m[3][3]; double& operator()(int i, int j) { return m[i][j]; } } cov;
    for (int k = 0; k < 4; ++k) {
        PMScan s(*pm.m_SickScans[k]);
        memset(&cov, 0, sizeof cov);
        float r = csm.FMatchKeyFrame2(&s, cov);
        printf("kf %d %g %a %a %a", k, r, s.rx, s.ry, s.th);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) printf(" %a", cov(i, j));
        printf("\n");
    }
    csm.resetKeyFrame(pm.m_SickScans[0]);
    csm.setThreshold(0.5f, 0.2f);
    double p1[3] = {0.1, 0.1, 0.1}, p2[3] = {0.6, 0.0, 0.0}, p3[3] = {0.0, 0.0, 0.3};
    printf("thr %d %d %d %g %g\n", (int)csm.newKeyframeNeeded(p1), (int)csm.newKeyframeNeeded(p2), (int)csm.newKeyframeNeeded(p3),
           csm.threshold_dist_sq_, csm.threshold_angular_);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc >= 5 && !strcmp(argv[1], "csmx")) return run_csmx(argv[2], argv[3], argv[4]);
    if (argc >= 6 && !strcmp(argv[1], "accuracy")) {     // accuracy <laser> <log> <psm_out> <icp_out>
        CPolarMatch m(argv[2]);
        m.m_accuracy_psm_path = argv[4]; m.m_accuracy_icp_path = argv[5];
        m.testAccuracy(argv[3]);
        printf("failed %d %d\n", m.m_failed_psm, m.m_failed_icp);
        return 0;
    }
    if (argc >= 6 && !strcmp(argv[1], "run511")) {       // run511 <laser> <file> <traj_out> <n>
        CPolarMatch m(argv[2]);
        m.m_trajectory_path = argv[4];
        m.run511Data(argv[3], atoi(argv[5]));
        return 0;
    }
    if (argc >= 7 && !strcmp(argv[1], "corridor")) return run_corridor(argv[2], argv[3], argv[4], argv[5], argv[6]);
    if (argc >= 6 && !strcmp(argv[1], "readers")) return run_readers(argv[2], argv[3], argv[4], argv[5]);
    if (argc >= 6 && !strcmp(argv[1], "match")) return run_match(argv[2], argv[3], argv[4], atoi(argv[5]));
    if (argc >= 4 && !strcmp(argv[1], "pmap")) return run_pmap(argv[2], argv[3]);
    if (argc >= 7 && !strcmp(argv[1], "runlog")) {       // runlog <our|flirt> <laser> <log> <out> <n>
        CPolarMatch m(argv[3]);
        m.m_trajectory_path = argv[5];
        // (runOurFile passes an empty laser name to readCarmon, which then matches no record - as in the
        //  reference; the in-house log is therefore read with its tag and handed to the driver directly)
        if (!strcmp(argv[2], "our")) { if (!m.readCarmon(argv[4], argv[3])) return 2; m.runlogImproved(atoi(argv[6])); }
        else m.runFlirtFile(argv[4], atoi(argv[6]));
        return 0;
    }
    fprintf(stderr, "usage: host_dropin_test match <laser> <log> <tag> <n> | pmap <in> <out>\n");
    return 1;
}
