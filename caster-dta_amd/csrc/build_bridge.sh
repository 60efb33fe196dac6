#!/bin/bash
# Build lib/caster_gvp_torch.so: the C++ eager fast path (torch_bridge.cpp) -- host code only, g++ against the installed
# PyTorch's headers, linked to libcaster_gvp.so next to it.  Needs no GPU.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../lib"
PY=${PYTHON:-python3}
read -r TORCH_INC TORCH_LIB PY_INC CXX11 <<<"$($PY - <<'PYEOF'
import sysconfig, torch, os
from torch.utils import cpp_extension as ce
print(" ".join("-I" + p for p in ce.include_paths()).replace(" ", ";"), ce.library_paths()[0], sysconfig.get_paths()["include"],
      int(torch._C._GLIBCXX_USE_CXX11_ABI))
PYEOF
)"
TORCH_INC="${TORCH_INC//;/ }"
g++ -O2 -std=c++17 -fPIC -shared -Wall -Wno-unused-variable -o "$OUT/caster_gvp_torch.so.tmp" "$HERE/torch_bridge.cpp" \
  $TORCH_INC -I"$PY_INC" -I/opt/rocm/include -D__HIP_PLATFORM_AMD__=1 -DUSE_ROCM=1 -DTORCH_EXTENSION_NAME=caster_gvp_torch \
  -D_GLIBCXX_USE_CXX11_ABI=$CXX11 -DTORCH_API_INCLUDE_EXTENSION_H \
  -L"$TORCH_LIB" -ltorch -ltorch_cpu -ltorch_python -lc10 -lc10_hip -ltorch_hip -L"$OUT" -lcaster_gvp -L/opt/rocm/lib -lamdhip64 \
  -Wl,-rpath,'$ORIGIN' -Wl,-rpath,"$TORCH_LIB"
mv "$OUT/caster_gvp_torch.so.tmp" "$OUT/caster_gvp_torch.so"
