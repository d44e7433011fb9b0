"""INTEGRATION.md's replacement bodies compile (VERDICT r3 weak 12 / SURVEY.md 8b "Build caveat").

The reference's three classes are bound to liborbgpu through code blocks in INTEGRATION.md.  OpenCV 2.4, PCL and the
reference's own headers do not exist in this image, so the blocks marked `<!-- snippet: NAME -->` are extracted from
the document and compiled (g++ -c, templates instantiated) against tests/integration/cv_standin.hpp (cv::Mat,
cv::KeyPoint, cv::InputArray / OutputArray) and tests/integration/ref_standin.hpp (the members of ORBextractor, Frame,
MapPoint, KeyFrame, ORBmatcher, LoopClosing the blocks touch, with the reference's names).  Blocks sharing a NAME form
one translation unit, in document order."""
import collections
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PREAMBLE = {
    # the declarations INTEGRATION.md tells the maintainer to add to the reference's headers come from ref_standin.hpp
    "orbextractor": '#include "ref_standin.hpp"\n',
    "orbmatcher": '#include "ref_standin.hpp"\n',
    "pointcloudmapping": '#include "ref_standin.hpp"\n'
                         'namespace ORB_SLAM2 {\n'
                         'class PointCloudMapping {  // include/PointCloudMap.h:41-88 with the header change of section 3\n'
                         '  public:\n'
                         '    PointCloudMapping(double resolution_, LoopClosing *loopCloser_);\n'
                         '    ~PointCloudMapping();\n'
                         '    void insertKeyFrame(KeyFrame *kf);\n'
                         '    void shutdown();\n'
                         '  protected:\n'
                         '    double resolution = 0.04;\n'
                         '    LoopClosing *loopCloser = nullptr;\n'
                         '    struct Gpu;\n'
                         '    std::unique_ptr<Gpu> gpu;\n'
                         '};\n}\n',
    # a fragment of Tracking's member functions: wrapped in a function that declares the members it names
    "tracking_table": '#include "ref_standin.hpp"\n'
                      'using namespace ORB_SLAM2;\n'
                      'static const uint8_t *descRow(const Frame &F, int i) { return F.mDescriptors.ptr<uint8_t>(i); }\n'
                      'static const float *TcwOf(const Frame &F) { return F.mTcw.ptr<float>(); }\n'
                      'int tracking_fragment(Frame &mCurrentFrame, Frame &mLastFrame, std::vector<MapPoint *> &mvpLocalMapPoints,\n'
                      '                      orbgpu_shim::MapPointTableT<MapPoint> &gTable, orbgpu_shim::DeviceFrameT<Frame> &dCur,\n'
                      '                      orbgpu_shim::DeviceFrameT<Frame> &dLast, int mSensor, float th)\n'
                      '{\n'
                      '    orbgpu_shim::ORBmatcherT<Frame, MapPoint> matcher(0.9f, true);\n',
}
EPILOGUE = {"tracking_table": "    return nmatches + nToMatch;\n}\n"}
STATICS = ("float ORB_SLAM2::Frame::fx, ORB_SLAM2::Frame::fy, ORB_SLAM2::Frame::cx, ORB_SLAM2::Frame::cy, ORB_SLAM2::Frame::mnMinX, "
           "ORB_SLAM2::Frame::mnMaxX, ORB_SLAM2::Frame::mnMinY, ORB_SLAM2::Frame::mnMaxY, ORB_SLAM2::Frame::mfGridElementWidthInv, "
           "ORB_SLAM2::Frame::mfGridElementHeightInv;\n")


def snippets():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    units = collections.OrderedDict()
    for m in re.finditer(r"<!-- snippet: (\w+)[^>]*-->\s*```cpp\n(.*?)```", text, re.S):
        units.setdefault(m.group(1), []).append(m.group(2))
    return units


def test_the_document_marks_every_binding():
    units = snippets()
    assert set(units) == {"orbextractor", "orbmatcher", "tracking_table", "pointcloudmapping"}, list(units)
    assert len(units["orbmatcher"]) == 2  # the per-frame matchers and SearchByBoW


def test_integration_code_blocks_compile(tmp_path):
    flags = ["g++", "-std=c++17", "-O0", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter", "-Wno-unused-function",
             "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "orb_slam2_map_amd", "shim"),
             "-I" + os.path.join(ROOT, "tests", "integration"), "-c"]
    objs = []
    for name, blocks in snippets().items():
        src = tmp_path / (name + ".cc")
        src.write_text(PREAMBLE[name] + "\n".join(blocks) + EPILOGUE.get(name, "") + (STATICS if name == "orbmatcher" else ""))
        obj = str(tmp_path / (name + ".o"))
        r = subprocess.run(flags + [str(src), "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, "INTEGRATION.md block '%s' does not compile:\n%s" % (name, r.stdout[-4000:])
        objs.append(obj)
    # and they link against the library (every orbgpu_* symbol the shim templates call exists)
    main = tmp_path / "main.cc"
    main.write_text("int main() { return 0; }\n")
    pkg = os.path.join(ROOT, "orb_slam2_map_amd")
    r = subprocess.run(["g++", str(main)] + objs + ["-o", str(tmp_path / "linked"), "-L" + pkg, "-lorbgpu", "-Wl,-rpath," + pkg,
                        "-Wl,-rpath,/opt/rocm/lib", "-pthread"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-4000:]
