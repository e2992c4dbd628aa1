// MRPT-shaped facade, part 3 of 3 (<mrpt/gui.h> and the mrpt::opengl scene graph): src/pf_localization_main.cpp
// builds a 3-D scene of the particles every few steps and can show it in a window.  There is no display here
// (MRPT_HAS_WXWIDGETS is 0, so the application itself switches the window off): the scene graph keeps names and
// parameters so that the calls type-check and getByName / removeObject behave; saving a scene writes a short text
// inventory.  See <mrpt/base.h> for what this facade is and is not.
#ifndef LSA_MRPT_FACADE_GUI_H
#define LSA_MRPT_FACADE_GUI_H

#include "slam.h"

namespace mrpt {
namespace opengl {
using namespace mrpt::poses;
using namespace mrpt::utils;

enum TOpenGLFontStyle { FILL = 0, OUTLINE = 1, NICE = 2 };

class CRenderizable : public CSerializable {
public:
    void setName(const std::string& n) { m_name = n; }
    const std::string& getName() const { return m_name; }
    void setColor(double r, double g, double b, double a = 1.0) { m_col[0] = r; m_col[1] = g; m_col[2] = b; m_col[3] = a; }
    void setPose(const CPose2D& p) { m_pose = CPose3D(p); }
    void setPose(const CPose3D& p) { m_pose = p; }
    void setLocation(double x, double y, double z) { m_pose.x = x; m_pose.y = y; m_pose.z = z; }
    const CPose3D& getPose() const { return m_pose; }
    virtual const char* kind() const { return "object"; }
protected:
    std::string m_name;
    double m_col[4] = {1, 1, 1, 1};
    CPose3D m_pose;
};
typedef ptr_t<CRenderizable> CRenderizablePtr;

class CSetOfObjects : public CRenderizable {
public:
    static ptr_t<CSetOfObjects> Create() { return ptr_t<CSetOfObjects>(new CSetOfObjects); }
    void insert(const CRenderizablePtr& o) { m_objs.push_back(o); }
    size_t size() const { return m_objs.size(); }
    const char* kind() const override { return "set"; }
private:
    std::vector<CRenderizablePtr> m_objs;
};

class CDisk : public CRenderizable {
public:
    static ptr_t<CDisk> Create() { return ptr_t<CDisk>(new CDisk); }
    void setDiskRadius(float r_out, float r_in = 0) { m_rout = r_out; m_rin = r_in; }
    const char* kind() const override { return "disk"; }
private:
    float m_rout = 1, m_rin = 0;
};
typedef ptr_t<CDisk> CDiskPtr;

class CEllipsoid : public CRenderizable {
public:
    static ptr_t<CEllipsoid> Create() { return ptr_t<CEllipsoid>(new CEllipsoid); }
    void setLineWidth(float w) { m_lw = w; }
    void setQuantiles(float q) { m_q = q; }
    void set2DsegmentsCount(unsigned n) { m_seg = n; }
    template <class M>
    void setCovMatrix(const M& cov, int resize_to = -1)
    {
        const int n = resize_to > 0 ? resize_to : (int)cov.getRowCount();
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m_cov[i][j] = (i < n && j < n) ? cov(i, j) : 0.0;
    }
    const char* kind() const override { return "ellipsoid"; }
private:
    float m_lw = 1, m_q = 3;
    unsigned m_seg = 20;
    double m_cov[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
};
typedef ptr_t<CEllipsoid> CEllipsoidPtr;

class CPointCloud : public CRenderizable {
public:
    static ptr_t<CPointCloud> Create() { return ptr_t<CPointCloud>(new CPointCloud); }
    void enableColorFromZ(bool v) { m_z = v; }
    void setPointSize(float s) { m_ps = s; }
    void loadFromPointsMap(const slam::CSimplePointsMap* m) { if (m) { m_x = m->x; m_y = m->y; } }
    size_t size() const { return m_x.size(); }
    const char* kind() const override { return "point cloud"; }
private:
    bool m_z = false;
    float m_ps = 1;
    std::vector<float> m_x, m_y;
};
typedef ptr_t<CPointCloud> CPointCloudPtr;

class CCamera : public CRenderizable {
public:
    void setAzimuthDegrees(float v) { m_az = v; }
    void setElevationDegrees(float v) { m_el = v; }
    void setPointingAt(const CPose2D& p) { m_px = p.x(); m_py = p.y(); m_pz = 0; }
    void setPointingAt(double x, double y, double z) { m_px = x; m_py = y; m_pz = z; }
    void setZoomDistance(float z) { m_zoom = z; }
    void setOrthogonal(bool v = true) { m_ortho = v; }
private:
    float m_az = 0, m_el = 0, m_zoom = 1;
    double m_px = 0, m_py = 0, m_pz = 0;
    bool m_ortho = false;
};

class COpenGLViewport : public CSerializable {
public:
    CCamera& getCamera() { return m_cam; }
private:
    CCamera m_cam;
};
typedef ptr_t<COpenGLViewport> COpenGLViewportPtr;

class COpenGLScene : public CSerializable {
public:
    static ptr_t<COpenGLScene> Create() { return ptr_t<COpenGLScene>(new COpenGLScene); }
    void insert(const CRenderizablePtr& o) { if (o) m_objs.push_back(o); }
    CRenderizablePtr getByName(const std::string& name) const
    {
        for (size_t i = 0; i < m_objs.size(); ++i) if (m_objs[i]->getName() == name) return m_objs[i];
        return CRenderizablePtr();
    }
    void removeObject(const CRenderizablePtr& o)
    {
        for (size_t i = 0; i < m_objs.size(); ++i) if (m_objs[i].pointer() == o.pointer()) { m_objs.erase(m_objs.begin() + i); return; }
    }
    void enableFollowCamera(bool v) { m_follow = v; }
    COpenGLViewportPtr getViewport(const std::string& name = "main")
    {
        COpenGLViewportPtr& v = m_views[name];
        if (!v) v = COpenGLViewportPtr(new COpenGLViewport);
        return v;
    }
    COpenGLViewportPtr createViewport(const std::string& name) { return getViewport(name); }
    size_t size() const { return m_objs.size(); }
    void writeToFile(const std::string& file) const override;          // text inventory: kind, name, pose
private:
    std::vector<CRenderizablePtr> m_objs;
    std::map<std::string, COpenGLViewportPtr> m_views;
    bool m_follow = false;
};
typedef ptr_t<COpenGLScene> COpenGLScenePtr;

template <class T>
T* getAs(const CRenderizablePtr& o) { return dynamic_cast<T*>(o.pointer()); }

}  // namespace opengl

namespace gui {
using namespace mrpt::utils;
class CDisplayWindow3D : public CObject {
public:
    static ptr_t<CDisplayWindow3D> Create(const std::string& title = "", unsigned w = 400, unsigned h = 300)
    {
        (void)title; (void)w; (void)h;
        return ptr_t<CDisplayWindow3D>(new CDisplayWindow3D);
    }
    CDisplayWindow3D() : m_scene(opengl::COpenGLScene::Create()) {}
    void setCameraZoom(float) {}
    void setCameraAzimuthDeg(float) {}
    void setCameraElevationDeg(float) {}
    void setCameraPointingToPoint(float, float, float) {}
    opengl::COpenGLScenePtr& get3DSceneAndLock() { return m_scene; }
    void unlockAccess3DScene() {}
    void forceRepaint() {}
    int waitForKey() { return 0; }
    void addTextMessage(double, double, const std::string&, const TColorf& = TColorf(), const std::string& = "mono",
                        double = 10, opengl::TOpenGLFontStyle = opengl::FILL, size_t = 0) {}
private:
    opengl::COpenGLScenePtr m_scene;
};
typedef ptr_t<CDisplayWindow3D> CDisplayWindow3DPtr;
}  // namespace gui

}  // namespace mrpt

#endif
