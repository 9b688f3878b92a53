"""Build the ONLY native source of the reference -- mpiFFT4py/cython/maths.pyx
(transpose_Uc, dealias_filter, transpose_Umpi) -- from where it lies under
/root/reference into oracle/_ref/ (git-ignored; travels to the GPU box with the
snapshot).  Nothing of the reference is copied into the tracked tree: the .pyx
is read in place, cython writes its generated C file and gcc the extension
module into oracle/_ref/ only.

    python oracle/build_ref.py          # no-op when /root/reference is absent

The resulting module is test infrastructure: tests compare the oracle's
slab_unpack / apply_mask and the HIP mfft_slab_unpack / mfft_dealias_filter with
the reference's own compiled loops.
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
PYX = "/root/reference/mpiFFT4py/cython/maths.pyx"


def build(verbose=False):
    if not os.path.exists(PYX):
        return None
    import numpy as np
    os.makedirs(OUT, exist_ok=True)
    so = os.path.join(OUT, "ref_maths" + sysconfig.get_config_var("EXT_SUFFIX"))
    if os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(PYX):
        return so
    c_file = os.path.join(OUT, "ref_maths.c")
    # module name must match the file name: cython derives PyInit_<name> from --module-name
    subprocess.check_call([sys.executable, "-m", "cython", "-3", "--module-name", "ref_maths", PYX, "-o", c_file],
                          stdout=None if verbose else subprocess.DEVNULL, stderr=subprocess.STDOUT)
    inc = [sysconfig.get_paths()["include"], np.get_include()]
    cmd = ["gcc", "-O2", "-shared", "-fPIC", "-w", "-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION"]
    cmd += ["-I" + i for i in inc] + [c_file, "-o", so]
    subprocess.check_call(cmd)
    os.remove(c_file)            # keep only the binary
    return so


def load():
    """Import oracle/_ref/ref_maths if it has been built (returns None otherwise)."""
    import glob
    import importlib.util
    cands = glob.glob(os.path.join(OUT, "ref_maths*.so"))
    if not cands:
        return None
    spec = importlib.util.spec_from_file_location("ref_maths", cands[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build(verbose=True))
