// MRPT-shaped facade, part 2 of 3 (<mrpt/slam.h>): observations, actions, the log reader, metric maps and the
// particle filter as src/pf_localization_main.cpp drives them (see <mrpt/base.h> for what this is and is not).
//
//   CParticleFilter::executeOn(pdf, action, observations, stats)    (pf_localization_main.cpp:627-632)
//     prediction   lsa_pf_motion        particle (+) (odometry increment + N(0, sigma)), one draw per particle
//     update       lsa_score_poses_std  sum of log-likelihoods of the scan's end points in the likelihood field
//     ESS, resampling  lsa_pose_argmax, lsa_pf_weights, lsa_pf_resample (systematic), lsa_gather_poses
//   The particles live on the GPU; a host mirror serves getParticlePose / getW / getMean / getCovariance.
#ifndef LSA_MRPT_FACADE_SLAM_H
#define LSA_MRPT_FACADE_SLAM_H

#include "base.h"

namespace mrpt {
namespace opengl {
class CSetOfObjects;
typedef utils::ptr_t<CSetOfObjects> CSetOfObjectsPtr;
CSetOfObjectsPtr lsa_make_set_of_objects();
}

namespace slam {
using namespace mrpt::poses;
using namespace mrpt::utils;
using mrpt::system::TTimeStamp;

// ---- observations -------------------------------------------------------------------------------------------------
class CObservation : public CSerializable {
public:
    TTimeStamp timestamp = 0;
    std::string sensorLabel;
};
typedef ptr_t<CObservation> CObservationPtr;

class CObservation2DRangeScan : public CObservation {
public:
    std::vector<float> scan;              // ranges [m]
    std::vector<char> validRange;
    float aperture = 3.14159265f;         // field of view [rad]
    bool rightToLeft = true;
    float maxRange = 80.0f;
    float startAngle = -1.5707963f;       // bearing of scan[0] in the sensor frame (CARMEN's start_angle)
    CPose3D sensorPose;
    static ptr_t<CObservation2DRangeScan> Create() { return ptr_t<CObservation2DRangeScan>(new CObservation2DRangeScan); }
};
typedef ptr_t<CObservation2DRangeScan> CObservation2DRangeScanPtr;

class CSimplePointsMap : public CSerializable {
public:
    std::vector<float> x, y;
    void clear() { x.clear(); y.clear(); }
    size_t size() const { return x.size(); }
    void insertPoint(float px, float py) { x.push_back(px); y.push_back(py); }
};

class CSensoryFrame : public CSerializable {
public:
    typedef std::vector<CObservationPtr>::iterator iterator;
    static ptr_t<CSensoryFrame> Create() { return ptr_t<CSensoryFrame>(new CSensoryFrame); }
    void insert(const CObservationPtr& o) { m_obs.push_back(o); }
    size_t size() const { return m_obs.size(); }
    CObservationPtr getObservationByIndex(size_t i) const { return m_obs.at(i); }
    iterator begin() { return m_obs.begin(); }
    iterator end() { return m_obs.end(); }
    bool insertObservationsInto(CSimplePointsMap* map) const;          // end points of the range scans, sensor frame
private:
    std::vector<CObservationPtr> m_obs;
};
typedef ptr_t<CSensoryFrame> CSensoryFramePtr;

// ---- actions ------------------------------------------------------------------------------------------------------
class CAction : public CSerializable {};
class CActionRobotMovement2D : public CAction {
public:
    enum TDrawSampleMotionModel { mmGaussian = 0, mmThrun };
    struct TMotionModelOptions {
        TDrawSampleMotionModel modelSelection = mmGaussian;
        struct TOptions_GaussianModel {
            float a1 = 0.01f, a2 = 0.001f, a3 = 0.0174533f, a4 = 0.05f, minStdXY = 0.01f, minStdPHI = 0.0034907f;
        } gausianModel;
    };
    CPosePDFPtr poseChange;
    TMotionModelOptions motionModelConfiguration;
    CActionRobotMovement2D() : poseChange(new CPosePDF) {}
    CActionRobotMovement2D(const CActionRobotMovement2D& o) : CAction(o), poseChange(new CPosePDF(*o.poseChange)),
                                                              motionModelConfiguration(o.motionModelConfiguration) {}
    // Gaussian odometry model: mean = the increment, sigma_xy = minStdXY + a1 |d| + a2 |dphi|, sigma_phi = minStdPHI + a3 |d| + a4 |dphi|
    void computeFromOdometry(const CPose2D& increment, const TMotionModelOptions& opts);
};
typedef ptr_t<CActionRobotMovement2D> CActionRobotMovement2DPtr;

class CActionCollection : public CSerializable {
public:
    static ptr_t<CActionCollection> Create() { return ptr_t<CActionCollection>(new CActionCollection); }
    void insert(const CActionRobotMovement2D& a) { m_act.push_back(CActionRobotMovement2DPtr(new CActionRobotMovement2D(a))); }
    size_t size() const { return m_act.size(); }
    CActionRobotMovement2DPtr getBestMovementEstimation() const { return m_act.empty() ? CActionRobotMovement2DPtr() : m_act[0]; }
private:
    std::vector<CActionRobotMovement2DPtr> m_act;
};
typedef ptr_t<CActionCollection> CActionCollectionPtr;

// ---- the log: a CARMEN log here (ROBOTLASER1 / FLASER lines), one <action, sensory frame> pair per laser line --------
class CRawlog {
public:
    static bool getActionObservationPairOrObservation(CFileGZInputStream& in, CActionCollectionPtr& action,
                                                      CSensoryFramePtr& observations, CObservationPtr& observation,
                                                      size_t& rawlogEntry);
};
// one laser line -> scan + laser pose + odometry pose; false if the line is not a laser line
bool lsa_parse_carmen_laser(const std::string& line, CObservation2DRangeScan& scan, double laser_pose[3], double odo_pose[3]);

// ---- maps ---------------------------------------------------------------------------------------------------------
class CSimpleMap : public CSerializable {                  // poses + scans: the laser lines of a CARMEN log
public:
    struct Entry { CPose2D pose; CObservation2DRangeScanPtr scan; };
    std::vector<Entry> entries;
    size_t size() const { return entries.size(); }
    void readFromFile(const std::string& file) override;
};

struct TMetricMapInitializer {
    float min_x = -10, max_x = 10, min_y = -10, max_y = 10, resolution = 0.10f;
    bool extent_given = false;
};
class TSetOfMetricMapInitializers {
public:
    std::vector<TMetricMapInitializer> grids;
    void loadFromConfigFile(const CConfigFileBase& cfg, const std::string& section);   // occupancyGrid_count, occupancyGrid_##_*
    void dumpToConsole() const;
};

struct lsa_gridmap_impl;
class COccupancyGridMap2D : public CSerializable {
public:
    struct TEntropyInfo { double H = 0, I = 0, mean_H = 0, mean_I = 0, effectiveMappedArea = 0; unsigned long effectiveMappedCells = 0; };
    COccupancyGridMap2D(float min_x = -10, float max_x = 10, float min_y = -10, float max_y = 10, float resolution = 0.1f);
    ~COccupancyGridMap2D();
    void setSize(float min_x, float max_x, float min_y, float max_y, float resolution);
    float getXMin() const { return m_xmin; }
    float getXMax() const { return m_xmax; }
    float getYMin() const { return m_ymin; }
    float getYMax() const { return m_ymax; }
    float getResolution() const { return m_res; }
    unsigned getSizeX() const { return m_sx; }
    unsigned getSizeY() const { return m_sy; }
    // p(free) of a cell, MRPT's convention (1 = free, 0 = occupied, 0.5 = unknown)
    float getCell(int cx, int cy) const { return (cx < 0 || cy < 0 || cx >= (int)m_sx || cy >= (int)m_sy) ? 0.5f : 1.0f - m_occ[(size_t)cx * m_sy + cy]; }
    int x2idx(float x) const { return (int)std::floor((x - m_xmin) / m_res); }
    int y2idx(float y) const { return (int)std::floor((y - m_ymin) / m_res); }
    void computeEntropy(TEntropyInfo& info) const;
    bool saveAsBitmapFile(const std::string& file) const;              // binary PGM + "# limits" comment whatever the extension
    void readFromFile(const std::string& file) override;               // the same format
    void getAs3DObject(opengl::CSetOfObjectsPtr& out) const;
    // builds the map on the GPU (CPMap's counting grid, lsa_grid_*) from poses + scans
    void buildFromSimpleMap(const CSimpleMap& sm, bool fit_extent);
    // [x][y] occupancy probabilities, 0.5 = unknown
    const std::vector<float>& occupancy() const { return m_occ; }
    // likelihood field on the GPU, built on first use
    lsa_gridmap_impl* device() const;
private:
    float m_xmin, m_xmax, m_ymin, m_ymax, m_res;
    unsigned m_sx, m_sy;
    std::vector<float> m_occ;
    mutable lsa_gridmap_impl* m_dev = nullptr;
};
typedef ptr_t<COccupancyGridMap2D> COccupancyGridMap2DPtr;

class CLandmarksMap : public CSerializable {
public:
    bool saveToMATLABScript2D(const std::string& file) const { (void)file; return false; }
};
typedef ptr_t<CLandmarksMap> CLandmarksMapPtr;

class CMultiMetricMap : public CSerializable {
public:
    std::vector<COccupancyGridMap2DPtr> m_gridMaps;
    CLandmarksMapPtr m_landmarksMap;
    void setListOfMaps(const TSetOfMetricMapInitializers* init);
    void loadFromProbabilisticPosesAndObservations(const CSimpleMap& sm);
private:
    std::vector<bool> m_fit;
};

// ---- particle filter ----------------------------------------------------------------------------------------------
struct TKLDParams {
    double KLD_binSize_XY = 0.2, KLD_binSize_PHI = 0.1745, KLD_delta = 0.01, KLD_epsilon = 0.02;
    unsigned KLD_minSampleSize = 250, KLD_maxSampleSize = 100000;
    void loadFromConfigFile(const CConfigFileBase& cfg, const std::string& section);
};
struct TMonteCarloLocalizationParams {
    CMultiMetricMap* metricMap = nullptr;
    TKLDParams KLD_params;
};

struct lsa_mcl_impl;
class CMonteCarloLocalization2D : public CSerializable {
public:
    explicit CMonteCarloLocalization2D(size_t M = 1);
    ~CMonteCarloLocalization2D();
    TMonteCarloLocalizationParams options;
    void resetUniformFreeSpace(COccupancyGridMap2D* map, double freeCellsThreshold = 0.7, int particlesCount = -1,
                               double x_min = -1e10, double x_max = 1e10, double y_min = -1e10, double y_max = 1e10,
                               double phi_min = -3.14159265358979323846, double phi_max = 3.14159265358979323846);
    void resetUniform(double x_min, double x_max, double y_min, double y_max, double phi_min = -3.14159265358979323846,
                      double phi_max = 3.14159265358979323846, int particlesCount = -1);
    size_t size() const;
    size_t particlesCount() const { return size(); }
    CPose2D getParticlePose(size_t i) const;
    double getW(size_t i) const;                                       // log weight
    void getMean(CPose2D& mean) const;
    void getCovarianceAndMean(math::CMatrixDouble33& cov, CPose2D& mean) const;
    void getCovariance(math::CMatrixDouble& cov) const;
    double ESS() const;                                                // in (0, 1]
    bool saveToTextFile(const std::string& file) const;               // x y phi log_w per line
    template <class T>
    T getAs3DObject() const { return T(opengl::lsa_make_set_of_objects()); }
    lsa_mcl_impl* impl() const { return m_impl; }
private:
    CMonteCarloLocalization2D(const CMonteCarloLocalization2D&);
    CMonteCarloLocalization2D& operator=(const CMonteCarloLocalization2D&);
    lsa_mcl_impl* m_impl;
};

class CParticleFilter {
public:
    enum TParticleFilterAlgorithm { pfStandardProposal = 0, pfAuxiliaryPFStandard, pfOptimalProposal, pfAuxiliaryPFOptimal };
    enum TParticleResamplingAlgorithm { prMultinomial = 0, prResidual, prStratified, prSystematic };
    struct TParticleFilterOptions {
        bool adaptiveSampleSize = false;
        double BETA = 0.5;
        unsigned sampleSize = 1;
        double powFactor = 1.0;
        int PF_algorithm = pfStandardProposal;
        int resamplingMethod = prSystematic;                           // the scheme run here whatever the file asks for
        double max_loglikelihood_dyn_range = 15;
        void loadFromConfigFile(const CConfigFileBase& cfg, const std::string& section);
        void dumpToConsole() const;
    };
    struct TParticleFilterStats {
        double ESS_beforeResample = 0, weightsVariance_beforeResample = 0;
    };
    TParticleFilterOptions m_options;
    void executeOn(CMonteCarloLocalization2D& pdf, const CActionCollection* action, const CSensoryFrame* observation,
                   TParticleFilterStats* stats = nullptr);
};

}  // namespace slam
}  // namespace mrpt

#endif
