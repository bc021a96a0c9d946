"""Kernel plugins: compile a user's functor into a shared library that libl3k.so loads at run time.

The reference instantiates its element loops on the type of the user's lambda, at the application's compile time.  The
device path needs the same thing for gfx950: `compile_kernel` writes a translation unit with the functor and the
requested shapes, runs hipcc on it (in-tree cache keyed by the content) and loads the result with l3k_plugin_load; the
new kernel id is then usable with MatrixFreeSystem / BoundaryTerm / integrate like the built-in ones.

    from l3ster_amd import plugin
    kid = plugin.compile_kernel("MyDiffusion", '''
        struct MyDiffusion {
            static constexpr l3k::KernelParams params{.dimension = 3, .n_equations = 7, .n_unknowns = 4};
            double k = 1., s = 1.;
            template <typename In, typename Out> L3K_HD void operator()(const In& in, Out& out) const {
                auto& [operators, rhs] = out;  auto& [A0, Ax, Ay, Az] = operators;
                Ax(0, 1) = -k; /* ... the body of the reference's lambda, unchanged ... */
            }
        };''', kernel_id=1000, shapes=[(2, 3, 1), (4, 5, 1)])
"""
import hashlib
import os
import shutil
import subprocess

from . import build as _build
from . import capi

KIND = {"domain": 0, "boundary": 1, "residual": 2}
_loaded = {}


def compile_kernel(type_name, source, kernel_id, shapes, kind="domain", display_name=None, verbose=False, isa_scan=True):
    """shapes: (order, nq, ncols) for equation kernels, (order, nq) for residual kernels.  Returns kernel_id."""
    if kernel_id < 1000:
        raise capi.L3KError("plugin kernel ids start at 1000 (lower ids belong to the kernels compiled into libl3k.so)")
    k = KIND[kind]
    lines = ['#include "device/instantiate.hpp"', "namespace l3k::plugin", "{", source, "} // namespace l3k::plugin"]
    T = f"::l3k::plugin::{type_name}"
    spec = "ResidualId" if k == 2 else "KernelId"
    lines += [f"template <> struct l3k::dev::{spec}< {T} > {{ static constexpr int value = {kernel_id}; }};",
              f'L3K_PLUGIN_KERNEL({kernel_id}, {k}, {T}, "{display_name or type_name}")']
    for sh in shapes:
        if k == 0:
            lines.append(f"L3K_INSTANTIATE({T}, {sh[0]}, {sh[1]}, {sh[2]})")
        elif k == 1:
            lines.append(f"L3K_INSTANTIATE_BOUNDARY({T}, {sh[0]}, {sh[1]}, {sh[2]})")
        else:
            lines.append(f"L3K_INSTANTIATE_RESIDUAL({T}, {sh[0]}, {sh[1]})")
    text = "\n".join(lines) + "\n"
    key = hashlib.sha256((text + _build._headers_digest()).encode()).hexdigest()[:20]
    if key in _loaded:
        return kernel_id
    out_dir = os.path.join(_build.HERE, "_build", "plugins")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, f"libl3k_plugin_{type_name}_{key}.so")
    if not os.path.exists(so):
        src = os.path.join(out_dir, f"plugin_{type_name}_{key}.hip")
        open(src, "w").write(text)
        libdir = os.path.dirname(capi.LIB_PATH)
        libname = os.path.basename(capi.LIB_PATH)[3:-3]
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        cmd = [hipcc] + _build.COMMON + _build.DEVICE + ["-shared", src, "-o", so + ".tmp", f"-L{libdir}", f"-l{libname}",
                                                          f"-Wl,-rpath,{libdir}"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            if os.path.exists(so + ".tmp"):
                os.unlink(so + ".tmp")
            raise capi.L3KError(f"kernel plugin failed to compile:\n{r.stderr[-4000:]}")
        # the assembly kernels of a plugin carry the library's hand-written DPP instructions: the same ISA scan as for libl3k.so,
        # BEFORE the library is loaded.  A plugin that cannot be scanned (no gfx950 code object found, a compressed bundle, no
        # llvm-objdump) is refused unless the caller waived the scan (isa_scan=False)
        if isa_scan:
            from . import isa_check
            try:
                report = isa_check.scan(so + ".tmp")
            except isa_check.IsaScanError as exc:
                os.unlink(so + ".tmp")
                raise capi.L3KError(f"kernel plugin: the DPP hazard scan could not run ({exc}); refusing to load an unscanned plugin "
                                    "(compile_kernel(..., isa_scan=False) loads it anyway)") from None
            if report["hazards"]:
                os.unlink(so + ".tmp")
                raise capi.L3KError(f"kernel plugin: DPP hazard in the generated code, refusing to load it: {report['hazards'][:3]}")
            if verbose:
                print(f"[l3k plugin] ISA scan: {report['n_code_objects']} gfx950 code object(s), {report['n_dpp']} DPP instructions, no hazard")
        os.replace(so + ".tmp", so)
        if verbose:
            print(f"[l3k plugin] built {so}")
    capi.check(capi.load().l3k_plugin_load(so.encode()))
    _loaded[key] = so
    return kernel_id
