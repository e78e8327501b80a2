"""g++ command of `make torch_ext`: compiles eslam_torch_ext.cpp against the installed torch (headers + libtorch) into a Python
extension module that links against libeslam_hip.so in the same directory.  No device code, no hipcc."""
import os
import subprocess
import sys
import sysconfig

import torch
from torch.utils import cpp_extension as ce

src, out = sys.argv[1], sys.argv[2]
libdir = os.path.dirname(os.path.abspath(out))
tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
inc = ce.include_paths(device_type="cuda") + ["/opt/rocm/include", sysconfig.get_paths()["include"]]
cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
       "-D_GLIBCXX_USE_CXX11_ABI=" + str(int(torch._C._GLIBCXX_USE_CXX11_ABI)), "-DTORCH_EXTENSION_NAME=eslam_torch_ext",
       "-DTORCH_API_INCLUDE_EXTENSION_H"] + ["-I" + i for i in inc] + [src, "-o", out, "-L" + tlib, "-L" + libdir, "-leslam_hip",
       "-ltorch", "-ltorch_cpu", "-ltorch_python", "-lc10", "-lc10_hip", "-ltorch_hip", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + tlib]
print(" ".join(cmd))
sys.exit(subprocess.call(cmd))
