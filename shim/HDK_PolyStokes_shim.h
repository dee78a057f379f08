// Houdini-side shim of the MI355X PolyStokes library: the ONLY translation unit that includes HDK headers.
// It registers the same DOP as the reference plugin (exec/HDK_PolyStokes.h:113-117, exec/HDK_PolyStokes.C:26-31,210-216)
// with the same parameter tokens, so existing .hipnc files (scenes/jelly_jam/*) load unchanged, and forwards
// solveGasSubclass() to the C ABI of include/polystokes.h.  Built only where a Houdini toolkit exists (shim/CMakeLists.txt
// needs $HFS); it cannot be compiled in the library's own build image and nothing in the library depends on it.
#ifndef HDK_POLYSTOKES_SHIM_H
#define HDK_POLYSTOKES_SHIM_H

#include <GAS/GAS_SubSolver.h>
#include <GAS/GAS_Utils.h>
#include <SIM/SIM_FieldUtils.h>
#include <SIM/SIM_Object.h>

#include <polystokes.h>

class HDK_PolyStokes : public GAS_SubSolver
{
public:
    // option accessors: the tokens are the reference's (exec/HDK_PolyStokes.h:23-43); note "minDensity"/"maxDensity" are read
    // with that capitalisation there although the parameters are "mindensity"/"maxdensity" — kept, it is what scenes rely on
    GET_DATA_FUNC_I("matrixSetup",                      MatrixSetup);
    GET_DATA_FUNC_I("solverType",                       SolverType);
    GET_DATA_FUNC_B("useInputSurfaceWeights",           UseInputSurfaceWeights);
    GET_DATA_FUNC_B("useInputCollisionWeights",         UseInputCollisionWeights);
    GET_DATA_FUNC_F("minDensity",                       MinDensity);
    GET_DATA_FUNC_F("maxDensity",                       MaxDensity);
    GET_DATA_FUNC_I("activeLiquidBoundaryLayerSize",    ActiveLiquidBoundaryLayerSize);
    GET_DATA_FUNC_I("activeSolidBoundaryLayerSize",     ActiveSolidBoundaryLayerSize);
    GET_DATA_FUNC_B("doReducedRegions",                 DoReducedRegions);
    GET_DATA_FUNC_B("doTile",                           DoTile);
    GET_DATA_FUNC_I("tileSize",                         TileSize);
    GET_DATA_FUNC_I("tilePadding",                      TilePadding);
    GET_DATA_FUNC_F(SIM_NAME_TOLERANCE,                 SolverTolerance);
    GET_DATA_FUNC_I("maxSolverIterations",              SolverMaxIterations);
    GET_DATA_FUNC_B("useWarmStart",                     UseWarmStart);
    GET_DATA_FUNC_B("exportMatrices",                   ExportMatrices);
    GET_DATA_FUNC_B("exportComponentMatrices",          ExportComponentMatrices);
    GET_DATA_FUNC_B("exportStats",                      ExportStats);
    GET_DATA_FUNC_B("doSolve",                          DoSolve);
    GET_DATA_FUNC_B("keepNonConvergedResults",          KeepNonConvergedResults);
    GET_DATA_FUNC_S("exportDataPrefix",                 ExportDataPrefix);
    // shim-only options (not in the reference; absent from old scenes, where they take these defaults)
    GET_DATA_FUNC_I("gpuDevice",                        GpuDevice);               // HIP device index, default 0
    GET_DATA_FUNC_B("hdkSampledWeights",                HdkSampledWeights);       // sample the 14 weight fields with HDK itself (1)
    GET_DATA_FUNC_B("debugGeometry",                    DebugGeometry);           // rebuild the reference's 36 debug point clouds (0)

protected:
    explicit HDK_PolyStokes(const SIM_DataFactory* factory);
    virtual ~HDK_PolyStokes();

    virtual bool solveGasSubclass(SIM_Engine& engine, SIM_Object* obj, SIM_Time time, SIM_Time timestep);

private:
    ps_context* myCtx;          // owns the device buffers; reused across substeps
    int         myCtxDevice;

    bool ensureContext(SIM_Object* obj);
    void publishDebugGeometry(SIM_Object* obj, const SIM_VectorField* velocity, fpreal dx);

    static const SIM_DopDescription* getDopDescription();

    DECLARE_STANDARD_GETCASTTOTYPE();
    DECLARE_DATAFACTORY(HDK_PolyStokes,
        GAS_SubSolver,
        "HDK PolyStokes Solver",
        getDopDescription());
};

#endif
