"""Host-only pieces of the reference-named facade (include/ellc_facade.hpp), exercised without a GPU: the text checkpoint
format of Frame.cpp:697-871 (default ostream float formatting, 6 significant digits, one blank after every value)."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROGRAM = r"""
#include "ellc_facade.hpp"
#include <cstdio>
int main(int argc, char** argv) {
  const std::string dir = argv[1];
  const int w = 5, h = 3;
  std::vector<float> in((size_t)w * h);
  FILE* f = std::fopen((dir + "/in.bin").c_str(), "rb");
  if (!f || std::fread(in.data(), 4, in.size(), f) != in.size()) return 2;
  std::fclose(f);
  ellc::text::writeMat(dir + "/mat.txt", in.data(), w, h);
  ellc::text::writeArray(dir + "/arr.txt", in.data(), in.size());
  std::vector<float> back(in.size(), -7.0f), back2(in.size(), -7.0f), shortread(in.size() + 3, -7.0f);
  ellc::text::readValues(dir + "/mat.txt", back.data(), back.size());
  ellc::text::readValues(dir + "/arr.txt", back2.data(), back2.size());
  ellc::text::readValues(dir + "/arr.txt", shortread.data(), shortread.size());   // file holds 3 values fewer
  f = std::fopen((dir + "/out.bin").c_str(), "wb");
  std::fwrite(back.data(), 4, back.size(), f);
  std::fwrite(back2.data(), 4, back2.size(), f);
  std::fwrite(shortread.data(), 4, shortread.size(), f);
  std::fclose(f);
  try {
    ellc::text::readValues(dir + "/missing.txt", back.data(), back.size());
    return 3;
  } catch (const std::runtime_error&) {
  }
  return 0;
}
"""


def _g(v):
    """default ostream << float: %g with 6 significant digits"""
    return "%g" % float(v)


def test_text_checkpoint_format_and_round_trip(tmp_path):
    vals = np.array([0.0, 1.0, -1.0, 0.1, 123456.789, 1234567.0, 1e-7, -2.5e-5, 3.14159274, 0.333333343, 1e10, 65.5, -0.0, 7.0, 0.5],
                    np.float32)
    vals.tofile(tmp_path / "in.bin")
    src = tmp_path / "t.cpp"
    src.write_text(PROGRAM)
    exe = tmp_path / "t"
    subprocess.run(["g++", "-std=c++11", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    subprocess.run([str(exe), str(tmp_path)], check=True)
    mat = (tmp_path / "mat.txt").read_text()
    arr = (tmp_path / "arr.txt").read_text()
    assert mat == "".join("".join(_g(v) + " " for v in vals[r * 5:(r + 1) * 5]) + "\n" for r in range(3))
    assert arr == "".join(_g(v) + " " for v in vals)
    out = np.fromfile(tmp_path / "out.bin", np.float32)
    six = np.array([float(_g(v)) for v in vals], np.float32)   # what 6 significant digits keep
    assert np.array_equal(out[:15], six) and np.array_equal(out[15:30], six)
    # a short file: the entries it does not hold are left as they were (the extraction fails in the stream's sentry)
    assert np.array_equal(out[30:45], six) and np.all(out[45:48] == -7.0)
