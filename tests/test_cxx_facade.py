"""CPU: the header-only C++ facade (include/caffe_facade.hpp: caffe::Net<float>, caffe::Blob<float>,
caffe::Caffe over the C ABI) compiles with g++, links against libdeepcut_hip.so and drives the host side of
the runtime (graph build, shapes, blobs, error on forward without a GPU) exactly like a tools/caffe.cpp-
style caller would."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include <cstdio>
#include <fstream>
#include "caffe_facade.hpp"
int main(int argc, char** argv) {
  using namespace caffe;
  Caffe::set_mode(Caffe::GPU);
  Net<float> net(argv[1], TEST);
  shared_ptr<Blob<float> > data = net.blob_by_name("data");
  data->Reshape(1, 3, 104, 136);
  net.Reshape();
  shared_ptr<Blob<float> > prob = net.blob_by_name("prob");
  std::printf("%s %d %d %d %d %d\n", net.name().c_str(), (int)net.layer_names().size(), (int)net.blob_names().size(),
              prob->channels(), prob->height(), prob->width());
  float* p = data->mutable_cpu_data();
  p[data->offset(0, 2, 5, 7)] = 3.f;
  std::printf("%d %g %d\n", data->count(), data->cpu_data()[data->offset(0, 2, 5, 7)], net.num_outputs());
  if (dc_device_count() == 0) {
    try { net.ForwardPrefilled(); std::printf("no error?\n"); return 1; }
    catch (const std::runtime_error& e) { std::printf("forward refused: %.20s\n", e.what()); }
  }
  try { Net<float> bad("/nonexistent.prototxt", TEST); return 1; }
  catch (const std::runtime_error& e) { std::printf("%.19s\n", e.what()); }
  return 0;
}
'''


def test_facade_compiles_links_and_runs(tmp_path):
    import caffe
    from deepcut_tools import deepercut_prototxt

    lib = caffe.lib_path()
    src = tmp_path / "facade_demo.cpp"
    src.write_text(SRC)
    proto = tmp_path / "net.prototxt"
    proto.write_text(deepercut_prototxt(152, 64, 64))
    exe = str(tmp_path / "facade_demo")
    subprocess.check_call(["g++", "-std=c++14", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src),
                           "-o", exe, lib, "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe, str(proto)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.strip().splitlines()
    assert lines[0] == "ResNet-152 734 332 14 13 17"
    assert lines[1] == "%d 3 3" % (3 * 104 * 136)
    assert lines[-1] == "Could not open file"
