"""The header-only C++ facade (include/caffe_facade.hpp: caffe::Net<float>, Blob<float>, Layer<float>, SyncedMemory,
LayerRegistry, Caffe over the C ABI) compiled with g++ and linked against libdeepcut_hip.so, driven like a
tools/caffe.cpp-style caller would.  CPU: host side (graph build, shapes, blobs, SyncedMemory on the host, Layer SetUp /
Reshape, the refusals).  -m gpu: ForwardPrefilled against the CPU oracle, Layer<float>::Forward (-> Forward_gpu) against
the oracle's layer functions, and the SyncedMemory state walk of the reference's own test
(src/caffe/test/test_syncedmem.cpp:16-125)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HOST_SRC = r'''
#include <cstdio>
#include <cstring>
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
  // SyncedMemory on the host (test_syncedmem.cpp:16-23,34-38,51-66)
  SyncedMemory mem(10);
  if (mem.head() != SyncedMemory::UNINITIALIZED || mem.size() != 10) return 2;
  if (!mem.cpu_data() || mem.head() != SyncedMemory::HEAD_AT_CPU) return 3;
  for (int round = 1; round <= 2; ++round) {
    void* c = mem.mutable_cpu_data();
    if (mem.head() != SyncedMemory::HEAD_AT_CPU) return 4;
    std::memset(c, round, mem.size());
    for (size_t i = 0; i < mem.size(); ++i) if (static_cast<const char*>(mem.cpu_data())[i] != round) return 5;
  }
  // Layer<float>: SetUp / Reshape on the host; Forward_cpu is refused (no CPU compute path)
  LayerParameter lp("name: \"c\" type: \"Convolution\" bottom: \"x\" top: \"y\" "
                    "convolution_param { num_output: 5 kernel_size: 3 stride: 2 pad: 1 }");
  shared_ptr<Layer<float> > layer = LayerRegistry<float>::CreateLayer(lp);
  Blob<float> x(2, 4, 9, 11), y;
  vector<Blob<float>*> bottom(1, &x), top(1, &y);
  layer->SetUp(bottom, top);
  std::printf("%s %s %d | %d %d %d %d | %d %d\n", lp.name().c_str(), layer->type(), (int)layer->blobs().size(), y.num(), y.channels(),
              y.height(), y.width(), layer->blobs()[0]->count(), layer->blobs()[1]->count());
  x.Reshape(1, 4, 16, 8);
  layer->Reshape(bottom, top);
  std::printf("%d %d %d %d\n", y.num(), y.channels(), y.height(), y.width());
  Caffe::set_mode(Caffe::CPU);
  try { layer->Forward(bottom, top); return 6; }
  catch (const std::runtime_error& e) { std::printf("%.18s\n", e.what()); }
  Blob<float> z(1, 4, 16, 8);
  x.mutable_cpu_data()[5] = 2.5f;
  z.CopyFrom(x);
  std::printf("%g %d\n", z.cpu_data()[5], (int)z.head());
  try { Net<float> bad("/nonexistent.prototxt", TEST); return 1; }
  catch (const std::runtime_error& e) { std::printf("%.19s\n", e.what()); }
  return 0;
}
'''


def _compile(tmp_path, name, src, hip=False):
    import caffe

    lib = caffe.lib_path()
    cpp = tmp_path / (name + ".cpp")
    cpp.write_text(src)
    exe = str(tmp_path / name)
    cmd = ["g++", "-std=c++14", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(cpp), "-o", exe, lib,
           "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib"]
    if hip:
        cmd += ["-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-L/opt/rocm/lib", "-lamdhip64", "-Wno-unused-result", "-Wno-deprecated-declarations"]
    subprocess.check_call(cmd)
    return exe


def test_facade_compiles_links_and_runs(tmp_path):
    from deepcut_tools import deepercut_prototxt

    proto = tmp_path / "net.prototxt"
    proto.write_text(deepercut_prototxt(152, 64, 64))
    exe = _compile(tmp_path, "facade_demo", HOST_SRC)
    out = subprocess.run([exe, str(proto)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.strip().splitlines()
    assert lines[0] == "ResNet-152 734 332 14 13 17"
    assert lines[1] == "%d 3 3" % (3 * 104 * 136)
    assert "c Convolution 2 | 2 5 5 6 | 180 5" in lines      # SetUp: weight [5,4,3,3] + bias [5]; (9+2-3)/2+1 = 5, (11+2-3)/2+1 = 6
    assert "1 5 8 4" in lines                                   # Reshape follows the bottom
    assert "Layer::Forward_cpu" in lines                        # CPU mode: refused, never silently computed elsewhere
    assert "2.5 1" in lines                                     # CopyFrom on the host: HEAD_AT_CPU
    assert lines[-1] == "Could not open file"


GPU_SRC = r'''
#include <cstdio>
#include <cstring>
#include <vector>
#include <hip/hip_runtime_api.h>
#include "caffe_facade.hpp"
using namespace caffe;
static int fail(int code, const char* what) { std::printf("FAIL %d %s\n", code, what); return code; }
static void dump(const char* path, const float* p, size_t n) { FILE* f = std::fopen(path, "wb"); std::fwrite(p, 4, n, f); std::fclose(f); }
static std::vector<float> slurp(const char* path) {
  FILE* f = std::fopen(path, "rb"); std::fseek(f, 0, SEEK_END); long n = std::ftell(f); std::fseek(f, 0, SEEK_SET);
  std::vector<float> v(n / 4); std::fread(v.data(), 4, v.size(), f); std::fclose(f); return v;
}
int main(int argc, char** argv) {
  const std::string dir = argv[3];
  Caffe::set_mode(Caffe::GPU);
  Caffe::SetDevice(0);
  // --- Net<float>: CopyTrainedLayersFrom + ForwardPrefilled (tools/caffe.cpp-style) --------------------------------
  {
    Net<float> net(argv[1], TEST);
    net.CopyTrainedLayersFrom(argv[2]);
    Blob<float>* data = net.input_blobs()[0];
    std::vector<float> img = slurp((dir + "/img.bin").c_str());
    if ((int)img.size() != data->count()) return fail(1, "input size");
    std::memcpy(data->mutable_cpu_data(), img.data(), img.size() * 4);
    const vector<Blob<float>*>& out = net.ForwardPrefilled();
    if (out.size() != 3) return fail(2, "outputs");
    for (size_t i = 0; i < 3; ++i) dump((dir + "/out" + std::to_string(i) + ".bin").c_str(), out[i]->cpu_data(), out[i]->count());
    if (net.blob_by_name("prob")->head() != SyncedMemory::SYNCED) return fail(3, "prob head after cpu_data");
  }
  // --- SyncedMemory state walk (test_syncedmem.cpp:25-31,40-47,68-123) -----------------------------------------------
  {
    SyncedMemory m0(10);
    if (!m0.cpu_data() || !m0.gpu_data() || !m0.mutable_cpu_data() || !m0.mutable_gpu_data()) return fail(10, "allocation");
    SyncedMemory m1(10);
    if (!m1.gpu_data() || m1.head() != SyncedMemory::HEAD_AT_GPU) return fail(11, "gpu_data on UNINITIALIZED -> HEAD_AT_GPU (syncedmem.cpp:52-58)");
    SyncedMemory mem(10);
    char back[10];
    for (int round = 1; round <= 2; ++round) {  // TestGPURead
      void* c = mem.mutable_cpu_data();
      if (mem.head() != SyncedMemory::HEAD_AT_CPU) return fail(12, "HEAD_AT_CPU");
      std::memset(c, round, mem.size());
      const void* g = mem.gpu_data();
      if (mem.head() != SyncedMemory::SYNCED) return fail(13, "SYNCED after gpu_data");
      if (hipMemcpy(back, g, 10, hipMemcpyDeviceToHost) != hipSuccess) return fail(14, "hipMemcpy");
      for (int i = 0; i < 10; ++i) if (back[i] != round) return fail(15, "device bytes");
    }
    SyncedMemory mw(10);
    for (int round = 1; round <= 2; ++round) {  // TestGPUWrite
      void* g = mw.mutable_gpu_data();
      if (mw.head() != SyncedMemory::HEAD_AT_GPU) return fail(16, "HEAD_AT_GPU");
      if (hipMemset(g, round, mw.size()) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return fail(17, "hipMemset");
      const char* c = static_cast<const char*>(mw.cpu_data());
      for (int i = 0; i < 10; ++i) if (c[i] != round) return fail(18, "host bytes");
      if (mw.head() != SyncedMemory::SYNCED) return fail(19, "SYNCED after cpu_data");
    }
  }
  // --- Layer<float>::Forward -> Forward_gpu: convolution with injected weights, then an in-place ReLU ----------------
  {
    LayerParameter lp("name: \"c\" type: \"Convolution\" bottom: \"x\" top: \"y\" convolution_param { num_output: 6 kernel_size: 3 stride: 2 pad: 1 }");
    shared_ptr<Layer<float> > conv = LayerRegistry<float>::CreateLayer(lp);
    Blob<float> x(2, 64, 9, 11), y;
    vector<Blob<float>*> bottom(1, &x), top(1, &y);
    conv->SetUp(bottom, top);
    std::vector<float> xv = slurp((dir + "/lx.bin").c_str()), wv = slurp((dir + "/lw.bin").c_str()), bv = slurp((dir + "/lb.bin").c_str());
    std::memcpy(x.mutable_cpu_data(), xv.data(), xv.size() * 4);
    std::memcpy(conv->blobs()[0]->mutable_cpu_data(), wv.data(), wv.size() * 4);
    std::memcpy(conv->blobs()[1]->mutable_cpu_data(), bv.data(), bv.size() * 4);
    conv->Forward(bottom, top);
    if (y.head() != SyncedMemory::HEAD_AT_GPU) return fail(20, "top stays on the device after Forward_gpu");
    shared_ptr<Layer<float> > relu = LayerRegistry<float>::CreateLayer(LayerParameter("name: \"r\" type: \"ReLU\" bottom: \"y\" top: \"y\""));
    vector<Blob<float>*> yy(1, &y);
    relu->SetUp(yy, yy);
    relu->Forward(yy, yy);  // in place, device to device
    dump((dir + "/ly.bin").c_str(), y.cpu_data(), y.count());
    std::printf("layer %d %d %d %d\n", y.num(), y.channels(), y.height(), y.width());
  }
  // --- Blob::Reshape beyond the device allocation of a blob that lives only on the device (blob.cpp:37-41) -----------
  {
    Blob<float> b(1, 2, 3, 4);
    float* g0 = b.mutable_gpu_data();
    if (!g0 || b.head() != SyncedMemory::HEAD_AT_GPU) return fail(30, "mutable_gpu_data");
    if (hipMemset(g0, 0x7f, b.count() * 4) != hipSuccess) return fail(31, "hipMemset");
    b.Reshape(4, 32, 24, 24);  // 73 728 floats: far beyond the first allocation
    if (b.head() == SyncedMemory::HEAD_AT_GPU) return fail(32, "grown blob must not keep the stale HEAD_AT_GPU image");
    float* g1 = b.mutable_gpu_data();
    if (!g1) return fail(33, "mutable_gpu_data after growth");
    if (hipMemset(g1, 0, (size_t)b.count() * 4) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return fail(34, "the new image holds count() floats");
    const float* c = b.cpu_data();  // reads count() floats back: out of bounds on the old allocation
    for (int i = 0; i < b.count(); i += 997) if (c[i] != 0.f) return fail(35, "host bytes after growth");
  }
  // --- layer->blobs()[0]->CopyFrom(a blob that lives on the device): the new weights must be the ones the forward uses ----
  {
    LayerParameter lp("name: \"c\" type: \"Convolution\" bottom: \"x\" top: \"y\" convolution_param { num_output: 6 kernel_size: 3 stride: 2 pad: 1 bias_term: false }");
    shared_ptr<Layer<float> > conv = LayerRegistry<float>::CreateLayer(lp);
    Blob<float> x(2, 64, 9, 11), y;
    vector<Blob<float>*> bottom(1, &x), top(1, &y);
    conv->SetUp(bottom, top);
    std::vector<float> xv = slurp((dir + "/lx.bin").c_str()), wv = slurp((dir + "/lw.bin").c_str());
    std::memcpy(x.mutable_cpu_data(), xv.data(), xv.size() * 4);
    std::memset(conv->blobs()[0]->mutable_cpu_data(), 0, wv.size() * 4);
    conv->Forward(bottom, top);  // all-zero filters: packs them, output 0
    if (y.cpu_data()[7] != 0.f) return fail(40, "zero filters");
    Blob<float> wsrc(6, 64, 3, 3);
    std::memcpy(wsrc.mutable_cpu_data(), wv.data(), wv.size() * 4);
    wsrc.gpu_data();
    float* wg = wsrc.mutable_gpu_data();  // HEAD_AT_GPU: the device image is the authoritative one
    (void)wg;
    if (wsrc.head() != SyncedMemory::HEAD_AT_GPU) return fail(41, "source on the device");
    conv->blobs()[0]->CopyFrom(wsrc);
    conv->Forward(bottom, top);
    dump((dir + "/ly2.bin").c_str(), y.cpu_data(), y.count());
  }
  std::printf("OK\n");
  return 0;
}
'''


@pytest.mark.gpu
def test_facade_on_the_gpu_matches_the_oracle(tmp_path, gpu_caffe, synth152):
    from deepcut_tools import deepercut_prototxt
    from oracle import oracle as O

    path, layers = synth152
    proto_text = deepercut_prototxt(152, 64, 64)
    proto = tmp_path / "net.prototxt"
    proto.write_text(proto_text)
    rs = np.random.RandomState(3)
    img = (rs.randn(1, 3, 64, 64) * 50).astype(np.float32)
    img.tofile(str(tmp_path / "img.bin"))
    lx = rs.randn(2, 64, 9, 11).astype(np.float32)
    lw = (rs.randn(6, 64, 3, 3) / 24.0).astype(np.float32)
    lb = rs.randn(6).astype(np.float32)
    for n, a in (("lx", lx), ("lw", lw), ("lb", lb)):
        a.tofile(str(tmp_path / (n + ".bin")))
    exe = _compile(tmp_path, "facade_gpu", GPU_SRC, hip=True)
    out = subprocess.run([exe, str(proto), path, str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout + out.stderr
    O.set_threads(min(16, os.cpu_count() or 1))
    ref = O.OracleNet(proto_text, layers).forward(data=img)
    for i, k in enumerate(["loc_pred", "next_pred", "prob"]):  # output order = std::set order (net.cpp:268-273)
        got = np.fromfile(str(tmp_path / ("out%d.bin" % i)), np.float32).reshape(ref[k].shape)
        assert float(np.abs(got - ref[k]).max()) <= 1e-3, k
    yref = O.relu_forward(O.conv_forward(lx, lw, lb, 2, 1, 1))
    assert "layer 2 6 5 6" in out.stdout
    got = np.fromfile(str(tmp_path / "ly.bin"), np.float32).reshape(yref.shape)
    assert float(np.abs(got - yref).max()) <= 1e-4
    assert (got >= 0).all() and (got == 0).any()
    # a parameter overwritten from a device-resident blob: the second forward runs with the copied filters
    y2ref = O.conv_forward(lx, lw, None, 2, 1, 1)
    got2 = np.fromfile(str(tmp_path / "ly2.bin"), np.float32).reshape(y2ref.shape)
    assert float(np.abs(got2 - y2ref).max()) <= 1e-4 and float(np.abs(got2).max()) > 0.1
