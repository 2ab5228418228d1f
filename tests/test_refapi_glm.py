"""include/eg3d_refapi_glm.hpp — the adapters between the reference's glm-typed structures and the shim's plain-float
mirrors — compiled against the REFERENCE TREE'S vendored glm and exercised on structures with the reference's member
names (tests/refapi/glm_adapter_check.cpp). Build container only: skipped where /root/reference is absent. Also: the
reference-exact call surface of include/eg3d_refapi.hpp (the argument lists of plg_matching_from_refpoints.hpp:53,55,
gauss_newton.hpp:20, outliers_filtering.hpp:18-21, the EdgeManager / PLGPConsensusManager base classes) must compile as
written in the reference's call sites — checked here without a GPU; tests/refapi/refapi_check.cpp RUNS them on one."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GLM = "/root/reference/external/glm"
PKG = os.path.join(ROOT, "edgegraph3d_amd")


@pytest.mark.skipif(not os.path.isdir(os.path.join(GLM, "glm")), reason="the reference tree is not present (build container only)")
def test_glm_adapters_compile_against_the_vendored_glm_and_round_trip(tmp_path):
    exe = str(tmp_path / "glm_adapter_check")
    subprocess.check_call(["g++", "-std=c++17", "-w", "-pthread", "-I", os.path.join(ROOT, "include"), "-I", GLM,
                           os.path.join(ROOT, "tests", "refapi", "glm_adapter_check.cpp"), "-L", PKG, "-leg3d", "-leg3d_host",
                           "-Wl,-rpath," + PKG, "-L", "/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "GLM-ADAPTER-OK" in out.stdout, out.stdout + out.stderr


def test_reference_call_sites_compile_verbatim(tmp_path):
    """pipelines.cpp:164, outliers_filtering.cpp:39 and edge_matcher.cpp:132 as the reference writes them, against the
    shim's types (syntax + overload resolution only; no GPU)."""
    src = r'''
#include "eg3d_refapi.hpp"
using namespace eg3d_ref;
using namespace std;
struct data_bundle { EdgeManager* em; PLGPConsensusManager* cm; };
vector<std::tuple<vec3, vector<PolyLineGraph2D::plg_point>, vector<int>>> run(const SfMData& sfmd, data_bundle* mfc, PLGMatchesManager& plgmm) {
  vector<std::tuple<vec3, vector<PolyLineGraph2D::plg_point>, vector<int>>> p3ds_r = plg_matching_from_refpoints_parallel(sfmd, mfc->em, mfc->cm, plgmm);
  vector<std::tuple<vec3, vector<PolyLineGraph2D::plg_point>, vector<int>>> serial = plg_matching_from_refpoints(sfmd, mfc->em, mfc->cm, plgmm);
  p3ds_r.insert(p3ds_r.end(), serial.begin(), serial.end());
  return p3ds_r;
}
void filters(SfMData& sfm_data_, int first_edgepoint, float gn_max_mse, int forced_min_filter) {
  vector<bool> inliers;
  gaussNewtonFiltering(sfm_data_, inliers, gn_max_mse);
  filter(sfm_data_, first_edgepoint);
  filter(sfm_data_, first_edgepoint, gn_max_mse);
  filter(sfm_data_, first_edgepoint, forced_min_filter);
  filter(sfm_data_, first_edgepoint, gn_max_mse, forced_min_filter);
}
int main() { return 0; }
'''
    p = tmp_path / "callsites.cpp"
    p.write_text(src)
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-pthread", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(p)])
