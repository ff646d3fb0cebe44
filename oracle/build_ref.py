"""TEST INFRASTRUCTURE: compile the reference's OWN CPU kernels into oracle/_ref/ref_C.so.

The sources are compiled from where they lie under /root/reference (never copied into this
repo); outputs go only to oracle/_ref/ (git-ignored, but shipped to the GPU box like any
other built .so).  What is built: csrc/vision.cpp (the pybind11 module with the reference's
14 bindings, vision.cpp:9-25) + csrc/cpu/ROIAlign_cpu.cpp + csrc/cpu/nms_cpu.cpp, CPU only
(no WITH_CUDA: the CUDA half needs nvcc and THC, which do not exist here -> unbuildable).

API drift: the two files call `AT_DISPATCH_FLOATING_TYPES(x.type(), ...)` (ROIAlign_cpu.cpp:242,
nms_cpu.cpp:71); torch 2.x requires `.scalar_type()` there.  That one token is rewritten in
the compiler's input stream (sed | g++ -x c++ -); no file of the reference is modified and
no stand-in header is written.
"""
import os
import subprocess
import sys
import sysconfig

REF = os.environ.get("OVIS_REFERENCE", "/root/reference")
CSRC = os.path.join(REF, "maskrcnn_benchmark", "csrc")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")


def main():
    if not os.path.isdir(CSRC):
        print("reference checkout not present; skipping oracle/_ref build")
        return 0
    from torch.utils import cpp_extension as ce
    import torch

    os.makedirs(OUT, exist_ok=True)
    inc = [f"-I{p}" for p in ce.include_paths()] + [f"-I{sysconfig.get_paths()['include']}", f"-I{CSRC}"]
    cxx11 = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    flags = ["-O2", "-fPIC", "-std=c++17", "-w", "-DTORCH_EXTENSION_NAME=ref_C",
             "-DTORCH_API_INCLUDE_EXTENSION_H", f"-D_GLIBCXX_USE_CXX11_ABI={cxx11}"]
    objs = []
    for rel, patch in (("vision.cpp", None),
                       ("cpu/ROIAlign_cpu.cpp", "s/AT_DISPATCH_FLOATING_TYPES(input\\.type(),/AT_DISPATCH_FLOATING_TYPES(input.scalar_type(),/"),
                       ("cpu/nms_cpu.cpp", "s/AT_DISPATCH_FLOATING_TYPES(dets\\.type(),/AT_DISPATCH_FLOATING_TYPES(dets.scalar_type(),/")):
        src = os.path.join(CSRC, rel)
        obj = os.path.join(OUT, rel.replace("/", "_") + ".o")
        if patch is None:
            subprocess.check_call(["g++", *flags, *inc, "-c", src, "-o", obj])
        else:
            sed = subprocess.Popen(["sed", patch, src], stdout=subprocess.PIPE)
            subprocess.check_call(["g++", *flags, *inc, "-x", "c++", "-", "-c", "-o", obj], stdin=sed.stdout)
            sed.wait()
        objs.append(obj)
    libdirs = ce.library_paths()
    link = ["g++", "-shared", "-o", os.path.join(OUT, "ref_C.so"), *objs]
    for d in libdirs:
        link += [f"-L{d}", f"-Wl,-rpath,{d}"]
    link += ["-lc10", "-ltorch_cpu", "-ltorch", "-ltorch_python"]
    subprocess.check_call(link)
    for o in objs:
        os.remove(o)
    print("built", os.path.join(OUT, "ref_C.so"))
    return 0


if __name__ == "__main__":
    sys.exit(main())
