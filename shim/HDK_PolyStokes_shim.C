// Houdini DSO entry of the MI355X PolyStokes drop-in (see HDK_PolyStokes_shim.h).  Replaces exec/HDK_PolyStokes.C of the
// reference: same hook (initializeSIM), same DOP ("hdk_polystokes", "HDK Polynomial Stokes Solver"), same parameter tokens,
// labels and defaults (exec/HDK_PolyStokes.C:88-216), same error messages (:251-314, :530-535, :597-604); everything between
// the field validation and the write-back is one call into libpolystokes_hip.so.
//
// NOT compiled in the library's build image (no HDK there): built by shim/CMakeLists.txt when $HFS is set.
#include "HDK_PolyStokes_shim.h"

#include <UT/UT_DSOVersion.h>
#include <UT/UT_Interrupt.h>
#include <UT/UT_VoxelArray.h>
#include <PRM/PRM_Include.h>
#include <SIM/SIM_PRMShared.h>
#include <SIM/SIM_DopDescription.h>
#include <SIM/SIM_GeometryCopy.h>
#include <SIM/SIM_ScalarField.h>
#include <SIM/SIM_VectorField.h>
#include <GU/GU_Detail.h>

#include <string>
#include <vector>

void initializeSIM(void*)
{
    IMPLEMENT_DATAFACTORY(HDK_PolyStokes);
}

HDK_PolyStokes::HDK_PolyStokes(const SIM_DataFactory* factory) : BaseClass(factory), myCtx(nullptr), myCtxDevice(-1) {}

HDK_PolyStokes::~HDK_PolyStokes()
{
    if (myCtx) ps_context_destroy(myCtx);
}

// ---- parameter interface ---------------------------------------------------------------------------------------------
// One row per parameter of the reference's template, in its order.  Kind: S string, F float, I int, T toggle, O ordinal menu.
namespace {
struct ParmRow { char kind; const char* token; const char* label; const char* sdef; double ndef; };
const ParmRow theRows[] = {
    {'S', GAS_NAME_VELOCITY,            "Velocity Field",                   "vel",              0},
    {'S', "valid",                      "Valid Field",                      "__valid",          0},
    {'S', "viscosity",                  "Viscosity Field",                  "viscosity",        0},
    {'S', GAS_NAME_DENSITY,             "Liquid Density Field",             "massdensity",      0},
    {'F', "mindensity",                 "Min Density",                      nullptr,            1},
    {'F', "maxdensity",                 "Max Density",                      nullptr,            100000},
    {'S', GAS_NAME_PRESSURE,            "Pressure Field",                   "pressure",         0},
    {'S', GAS_NAME_SURFACE,             "Surface Field",                    "surface",          0},
    {'S', GAS_NAME_COLLISION,           "Solid Collision Field",            "collision",        0},
    {'S', GAS_NAME_COLLISIONVELOCITY,   "Solid Collision Velocity Field",   "collisionvel",     0},
    {'S', "exportDataPrefix",           "Export Data Prefix",               "output_data/`opname(\"../..\")`.$FF.", 0},
    {'O', "matrixSetup",                "Matrix Setup",                     nullptr,            1},   // template ordinal 1 as in the reference (:187);
    {'O', "solverType",                 "Solver Type",                      nullptr,            1},   // scenes store the index-0 token
    {'T', "doSolve",                    "Do Solve",                         nullptr,            1},
    {'T', "keepNonConvergedResults",    "Keep Non-Converged Results",       nullptr,            1},
    {'T', "exportMatrices",             "Export Matrices",                  nullptr,            0},
    {'T', "exportComponentMatrices",    "Export Component Matrices",        nullptr,            0},
    {'T', "exportStats",                "Export Stats",                     nullptr,            0},
    {'T', "useWarmStart",               "Use Warm Start",                   nullptr,            1},
    {'F', SIM_NAME_TOLERANCE,           "Solver Tolerance",                 nullptr,            1e-3},
    {'F', "maxSolverIterations",        "Max Solver Iterations",            nullptr,            5000},
    {'T', "useInputSurfaceWeights",     "Use Input Surface Weights",        nullptr,            1},
    {'S', "surfaceweights",             "Surface Weights Field",            "surfaceweights",   0},
    {'T', "useInputCollisionWeights",   "Use Input Collision Weights",      nullptr,            1},
    {'S', "collisionweights",           "Collision Weights Field",          "collisionweights", 0},
    {'I', "activeLiquidBoundaryLayerSize", "Active Liquid Boundary Layer Size", nullptr,        2},
    {'I', "activeSolidBoundaryLayerSize",  "Active Solid Boundary Layer Size",  nullptr,        2},
    {'T', "doReducedRegions",           "Do Reduced Regions",               nullptr,            1},
    {'T', "doTile",                     "Do Tile",                          nullptr,            1},
    {'I', "tileSize",                   "Reduced Tile Size",                nullptr,            16},
    {'I', "tilePadding",                "Reduced Tile Padding",             nullptr,            2},
    // shim-only
    {'I', "gpuDevice",                  "GPU Device",                       nullptr,            0},
    {'T', "hdkSampledWeights",          "Sample Weights With HDK",          nullptr,            1},
    {'T', "debugGeometry",              "Publish Debug Geometry",           nullptr,            0},
};
constexpr int theRowCount = (int)(sizeof(theRows) / sizeof(theRows[0]));
}  // namespace

const SIM_DopDescription* HDK_PolyStokes::getDopDescription()
{
    static PRM_Name theMatrixChoices[] = { PRM_Name("pressurestress", "Pressure Stress SPD Form"), PRM_Name(0) };
    static PRM_Name theSolverChoices[] = { PRM_Name("pcg_matrix_vector_products", "Preconditioned CG - Factored Matrix Vector Products"), PRM_Name(0) };
    static PRM_ChoiceList theMatrixMenu(PRM_CHOICELIST_SINGLE, theMatrixChoices);
    static PRM_ChoiceList theSolverMenu(PRM_CHOICELIST_SINGLE, theSolverChoices);

    static PRM_Name     theNames[theRowCount];
    static PRM_Default  theDefaults[theRowCount];
    static PRM_Template theTemplates[theRowCount + 1];
    static bool built = false;
    if (!built) {
        for (int i = 0; i < theRowCount; ++i) {
            const ParmRow& r = theRows[i];
            theNames[i] = PRM_Name(r.token, r.label);
            theDefaults[i] = r.sdef ? PRM_Default(0, r.sdef) : PRM_Default(r.ndef);
            switch (r.kind) {
            case 'S': theTemplates[i] = PRM_Template(PRM_STRING, 1, &theNames[i], &theDefaults[i]); break;
            case 'F': {
                const bool density = std::string(r.token) == "mindensity" || std::string(r.token) == "maxdensity";
                theTemplates[i] = density ? PRM_Template(PRM_FLT, 1, &theNames[i], &theDefaults[i], 0, 0, 0, &PRM_SpareData::unitsDensity)
                                          : PRM_Template(PRM_FLT, 1, &theNames[i], &theDefaults[i]);
                break;
            }
            case 'I': theTemplates[i] = PRM_Template(PRM_INT, 1, &theNames[i], &theDefaults[i]); break;
            case 'T': theTemplates[i] = PRM_Template(PRM_TOGGLE, 1, &theNames[i], &theDefaults[i]); break;
            case 'O': theTemplates[i] = PRM_Template(PRM_ORD, 1, &theNames[i], &theDefaults[i],
                                                     std::string(r.token) == "matrixSetup" ? &theMatrixMenu : &theSolverMenu); break;
            }
        }
        theTemplates[theRowCount] = PRM_Template();
        built = true;
    }
    static SIM_DopDescription theDopDescription(true, "hdk_polystokes", "HDK Polynomial Stokes Solver", "$OS", classname(), theTemplates);
    setGasDescription(theDopDescription);
    return &theDopDescription;
}

// ---- dense <-> UT_VoxelArray ------------------------------------------------------------------------------------------
namespace {
// x-fastest dense copy of a raw field (the layout of ps_fields_in); flatten() walks the voxel tiles once
void toDense(const SIM_RawField& f, std::vector<float>& out)
{
    const UT_VoxelArrayF& a = *f.field();
    const exint nx = a.getXRes(), ny = a.getYRes(), nz = a.getZRes();
    out.resize((size_t)nx * ny * nz);
    a.flatten(out.data(), nx, nx * ny);
}
void fromDense(SIM_RawField& f, const std::vector<float>& in)
{
    UT_VoxelArrayF& a = *f.fieldNC();
    const exint nx = a.getXRes(), ny = a.getYRes();
    a.extractFromFlattened(in.data(), nx, nx * ny);
}
int interruptTrampoline(void* user) { return ((UT_Interrupt*)user)->opInterrupt() ? 1 : 0; }

// the 7 sample grids in the ABI's order (polystokes.h: center, faceX..Z, edgeYZ, edgeXZ, edgeXY)
const SIM_FieldSample theSamples[7] = { SIM_SAMPLE_CENTER, SIM_SAMPLE_FACEX, SIM_SAMPLE_FACEY, SIM_SAMPLE_FACEZ,
                                        SIM_SAMPLE_EDGEYZ, SIM_SAMPLE_EDGEXZ, SIM_SAMPLE_EDGEXY };
const char* const theSampleNames[7] = { "center", "faceX", "faceY", "faceZ", "edgeYZ", "edgeXZ", "edgeXY" };
}  // namespace

bool HDK_PolyStokes::ensureContext(SIM_Object* obj)
{
    const int dev = getGpuDevice();
    if (myCtx && myCtxDevice == dev) return true;
    if (myCtx) { ps_context_destroy(myCtx); myCtx = nullptr; }
    myCtx = ps_context_create(dev);
    myCtxDevice = dev;
    if (!myCtx) { addError(obj, SIM_MESSAGE, ps_last_error(nullptr), UT_ERROR_ABORT); return false; }
    return true;
}

bool HDK_PolyStokes::solveGasSubclass(SIM_Engine& engine, SIM_Object* obj, SIM_Time time, SIM_Time timestep)
{
    // ---- fields: the reference's fetches and checks, message for message (exec/HDK_PolyStokes.C:235-314) ----
    SIM_VectorField*       velocityField = getVectorField(obj, GAS_NAME_VELOCITY);
    SIM_VectorField*       validField = getVectorField(obj, "valid");
    const SIM_ScalarField* viscosityField = getScalarField(obj, "viscosity");
    const SIM_ScalarField* pressureField = getScalarField(obj, "pressure");
    const SIM_ScalarField* densityField = getScalarField(obj, "density");
    const SIM_ScalarField* surfaceField = getConstScalarField(obj, GAS_NAME_SURFACE);
    const SIM_VectorField* surfaceWeights = getVectorField(obj, "surfaceweights");
    const SIM_ScalarField* collisionField = getConstScalarField(obj, GAS_NAME_COLLISION);
    const SIM_VectorField* collisionWeights = getVectorField(obj, "collisionweights");
    const SIM_VectorField* collisionVelocityField = getConstVectorField(obj, GAS_NAME_COLLISIONVELOCITY);

    auto fail = [&](const char* msg, UT_ErrorSeverity sev) { addError(obj, SIM_MESSAGE, msg, sev); return false; };
    if (!velocityField) return fail("Velocity field is missing.", UT_ERROR_WARNING);
    if (!velocityField->isFaceSampled()) return fail("Velocity field must be a staggered grid.", UT_ERROR_ABORT);
    if (!validField) return fail("Valid field is missing.", UT_ERROR_ABORT);
    if (!validField->isAligned(velocityField)) return fail("Valid field must align with the velocity field.", UT_ERROR_ABORT);
    if (!surfaceField) return fail("Surface field is missing.", UT_ERROR_ABORT);
    if (!collisionField) return fail("Collision field is missing.", UT_ERROR_ABORT);
    if (!viscosityField) return fail("Viscosity field is missing.", UT_ERROR_ABORT);
    if (!pressureField) return fail("Pressure field is missing.", UT_ERROR_ABORT);
    if (!densityField) return fail("Density field is missing.", UT_ERROR_ABORT);
    fpreal32 constantLiquidDensity = 0.;
    if (!densityField->getField()->field()->isConstant(&constantLiquidDensity))
        return fail("Variable density is not currently supported", UT_ERROR_WARNING);
    if (!surfaceWeights && getUseInputSurfaceWeights())
        return fail("User requested to use input surface weights but that field is missing.", UT_ERROR_ABORT);
    if (!collisionWeights && getUseInputCollisionWeights())
        return fail("User requested to use input collision weights but that field is missing.", UT_ERROR_ABORT);
    if (!collisionVelocityField) return fail("Collision velocity field is missing.", UT_ERROR_ABORT);   // the reference dereferences it unchecked
    if (!ensureContext(obj)) return false;

    const fpreal dt = timestep;
    const fpreal dx = velocityField->getVoxelSize(0).maxComponent();

    // ---- parameters: one ps_params member per option (include/polystokes.h) ----
    ps_params p;
    ps_params_default(&p);
    p.mindensity = getMinDensity();                 p.maxdensity = getMaxDensity();
    // The two menus are read as the reference reads them (HDK_PolyStokes.h:23-24 -> units.h:76-94: the stored ordinal IS the enum value).  Each
    // menu has ONE entry (HDK_PolyStokes.C:150-168), so the UI — and every shipped scene — stores 0 = pressurestress / pcg_matrix_vector_products;
    // a hand-set ordinal 1 selects SolverType::EIGEN as it does in the reference (Solver.cpp:646-668) and the library runs it; any other matrix
    // scheme or solver comes back through ps_last_error as "Unsupported matrix setup." / "Unsupported Solver." (HDK_PolyStokes.C:530-535).
    p.matrixSetup = (int32_t)getMatrixSetup();
    p.solverType = (int32_t)getSolverType();
    p.doSolve = getDoSolve();                       p.keepNonConvergedResults = getKeepNonConvergedResults();
    p.exportMatrices = getExportMatrices();         p.exportComponentMatrices = getExportComponentMatrices();
    p.exportStats = getExportStats();               p.useWarmStart = getUseWarmStart();
    p.tolerance = getSolverTolerance();             p.maxSolverIterations = getSolverMaxIterations();
    p.useInputSurfaceWeights = getUseInputSurfaceWeights();
    p.useInputCollisionWeights = getUseInputCollisionWeights();
    p.activeLiquidBoundaryLayerSize = getActiveLiquidBoundaryLayerSize();
    p.activeSolidBoundaryLayerSize = getActiveSolidBoundaryLayerSize();
    p.doReducedRegions = getDoReducedRegions();     p.doTile = getDoTile();
    p.tileSize = getTileSize();                     p.tilePadding = getTilePadding();
    UT_String prefix;
    getExportDataPrefix(prefix);
    const std::string prefixStr = prefix.toStdString();
    p.exportDataPrefix = prefixStr.c_str();
    p.negateCollision = 0;      // the reference samples both SDFs with invert = false (Solver.cpp:304,322): pass the field as it is

    // ---- dense copies of the SIM fields ----
    std::vector<float> vel[3], cvel[3], valid[3], surf, coll, visc, w[14];
    for (int a = 0; a < 3; ++a) {
        toDense(*velocityField->getField(a), vel[a]);
        toDense(*collisionVelocityField->getField(a), cvel[a]);
        valid[a].resize(vel[a].size());
    }
    toDense(*surfaceField->getField(), surf);
    toDense(*collisionField->getField(), coll);
    toDense(*viscosityField->getField(), visc);

    ps_fields_in in = {};
    {
        int rx, ry, rz;
        surfaceField->getField()->getVoxelRes(rx, ry, rz);
        in.nx = rx; in.ny = ry; in.nz = rz;
    }
    in.dx = dx; in.dt = dt; in.density = constantLiquidDensity;   // fpreal (double) like HDK_PolyStokes.C:319-320
    const UT_Vector3 orig = velocityField->getOrig();
    in.orig[0] = orig.x(); in.orig[1] = orig.y(); in.orig[2] = orig.z();
    for (int a = 0; a < 3; ++a) { in.vel[a] = vel[a].data(); in.collisionvel[a] = cvel[a].data(); }
    in.surface = surf.data(); in.collision = coll.data(); in.viscosity = visc.data();

    if (getHdkSampledWeights()) {
        // Solver::buildIntegrationWeightsAlt (Solver.cpp:238-326) with HDK's own sampler: 7 liquid + 7 fluid volume-fraction
        // fields, 2 samples per axis, invert = false, min weight 0 — the library then classifies from exactly these numbers
        const UT_Vector3 size = velocityField->getSize();
        for (int s = 0; s < 7; ++s)
            for (int which = 0; which < 2; ++which) {
                SIM_RawField wf;
                wf.init(theSamples[s], orig, size, in.nx, in.ny, in.nz);
                wf.computeSDFWeightsSampled(which == 0 ? surfaceField->getField() : collisionField->getField(), 2, false, 0);
                toDense(wf, w[which * 7 + s]);
                in.weights[which * 7 + s] = w[which * 7 + s].data();
            }
    }

    ps_fields_out out = {};
    for (int a = 0; a < 3; ++a) { out.vel[a] = vel[a].data(); out.valid[a] = valid[a].data(); }

    UT_Interrupt* boss = UTgetInterrupt();
    ps_set_interrupt(myCtx, &interruptTrampoline, boss);

    ps_stats st;
    const int result = polystokes_step(myCtx, &p, &in, &out, &st);      // == HDK_PolyStokes::Solver::SolverResult (Solver.h:61-70)
    ps_set_interrupt(myCtx, nullptr, nullptr);

    if (result == PS_FAILED) return fail(ps_last_error(myCtx), UT_ERROR_ABORT);
    if (result == PS_UNSUPPORTED_SOLVER) return fail("Unsupported Solver.", UT_ERROR_ABORT);

    if (getDebugGeometry()) publishDebugGeometry(obj, velocityField, dx);

    // ---- write back (HDK_PolyStokes.C:556-606) ----
    for (int a = 0; a < 3; ++a) fromDense(*validField->getField(a), valid[a]);
    const bool keep = result == PS_SUCCESS || getKeepNonConvergedResults();
    if (keep) for (int a = 0; a < 3; ++a) fromDense(*velocityField->getField(a), vel[a]);
    if (getDoSolve()) {
        if (keep) { velocityField->pubHandleModification(); validField->pubHandleModification(); }
        else if (result == PS_NOCONVERGE) addError(obj, SIM_MESSAGE, "Solver did not converge, exiting...", UT_ERROR_ABORT);
        else addError(obj, SIM_MESSAGE, "Solver failed, exiting...", UT_ERROR_ABORT);
    }
    return result == PS_SUCCESS;
}

// The reference's printAllData (Solver.cpp:1030-1268) publishes labels / indices / weights as point clouds named after the
// options set in its constructor (exec/HDK_PolyStokes.C:36-78).  Same geometry names, same attributes ("pscale", "data"),
// filled from the library's arrays of the same names (ps_query_array / ps_read_array).
void HDK_PolyStokes::publishDebugGeometry(SIM_Object* obj, const SIM_VectorField* velocity, fpreal dx)
{
    static const char* const kinds[5] = { "Labels", "ReducedIndices", "ActiveIndices", "LiquidWeights", "FluidWeights" };
    const UT_Vector3 orig = velocity->getOrig();
    const UT_Vector3 size = velocity->getSize();
    int nx, ny, nz;
    {
        UT_Vector3I d = velocity->getDivisions();
        nx = d.x(); ny = d.y(); nz = d.z();
    }
    for (int s = 0; s < 7; ++s)
        for (int k = 0; k < 5; ++k) {
            const std::string name = std::string(theSampleNames[s]) + kinds[k];
            int32_t elem = 0;
            const int64_t n = ps_query_array(myCtx, name.c_str(), &elem);
            if (n <= 0) continue;
            std::vector<char> raw((size_t)n * elem);
            if (ps_read_array(myCtx, name.c_str(), raw.data(), (int64_t)raw.size()) != PS_SUCCESS) continue;
            SIM_RawField probe;                                   // only for indexToPos of this sample grid
            probe.init(theSamples[s], orig, size, nx, ny, nz);
            int rx, ry, rz;
            probe.getVoxelRes(rx, ry, rz);
            SIM_GeometryCopy* geo = getOrCreateGeometry(obj, name.c_str());
            SIM_GeometryAutoWriteLock lock(geo, SIM_DATA_ID_PRESERVE);
            GU_Detail* detail = &lock.getGdp();
            detail->clearAndDestroy();
            GA_RWHandleF pscale(detail->addFloatTuple(GA_ATTRIB_POINT, "pscale", 1, GA_Defaults(0)));
            GA_RWHandleF data(detail->addFloatTuple(GA_ATTRIB_POINT, "data", 1, GA_Defaults(-1.)));
            for (int z = 0; z < rz; ++z)
                for (int y = 0; y < ry; ++y)
                    for (int x = 0; x < rx; ++x) {
                        const size_t q = ((size_t)z * ry + y) * rx + x;
                        const float v = k < 3 ? (float)((const int32_t*)raw.data())[q] : ((const float*)raw.data())[q];
                        if (k < 3 && v < 0) continue;             // unassigned entries carry no point (as in the reference)
                        UT_Vector3 pos;
                        probe.indexToPos(x, y, z, pos);
                        const GA_Offset pt = detail->appendPoint();
                        detail->setPos3(pt, pos);
                        pscale.set(pt, (float)(0.25 * dx));
                        data.set(pt, v);
                    }
        }
    // region centres of mass
    int32_t elem = 0;
    const int64_t n = ps_query_array(myCtx, "reducedRegionCenterOfMass", &elem);
    if (n > 0) {
        std::vector<double> com((size_t)n);
        if (ps_read_array(myCtx, "reducedRegionCenterOfMass", com.data(), n * 8) == PS_SUCCESS) {
            SIM_GeometryCopy* geo = getOrCreateGeometry(obj, "reducedRegionCenterOfMass");
            SIM_GeometryAutoWriteLock lock(geo, SIM_DATA_ID_PRESERVE);
            GU_Detail* detail = &lock.getGdp();
            detail->clearAndDestroy();
            GA_RWHandleF data(detail->addFloatTuple(GA_ATTRIB_POINT, "data", 1, GA_Defaults(-1.)));
            for (int64_t r = 0; r < n / 3; ++r) {
                const GA_Offset pt = detail->appendPoint();
                // basis-frame coordinates (cell index * dx) + the grid origin + half a cell
                detail->setPos3(pt, UT_Vector3((float)(com[(size_t)r * 3] + orig.x() + 0.5 * dx), (float)(com[(size_t)r * 3 + 1] + orig.y() + 0.5 * dx),
                                               (float)(com[(size_t)r * 3 + 2] + orig.z() + 0.5 * dx)));
                data.set(pt, (float)r);
            }
        }
    }
}
