"""MFMA instructions per kernel in the compiled ISA (caster-dta_amd/lib/_obj/*-gfx950.s, kept by build.sh)."""
import glob, os, re, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = sys.argv[1] if len(sys.argv) > 1 else r"conv_bwd_kernelILi1|conv_quad_kernelILi1ELi[12]ELi[12]|node_bwd|edge_bwd_kernelILi1|embed_bwd_kernelILi20|head_bwd|node_quad_kernelILb1"
for path in sorted(glob.glob(os.path.join(REPO, "caster-dta_amd", "lib", "_obj", "gvp_quad*-hip-amdgcn-amd-amdhsa-gfx950.s"))):
    txt = open(path).read()
    parts = re.split(r"^(_Z[A-Za-z0-9_]+):[^\n]*\n", txt, flags=re.M)
    for i in range(1, len(parts) - 1, 2):
        name, body = parts[i], parts[i + 1].split("s_endpgm")[0]
        if re.search(pat, name):
            f32 = len(re.findall(r"v_mfma_f32_16x16x4_f32", body))
            bf = len(re.findall(r"v_mfma_f32_16x16x16_bf16", body))
            print(f"{name[22:74]:54s} fp32 16x16x4 {f32:4d}   bf16 16x16x16 {bf:4d}   cvt_pk {len(re.findall('v_cvt_pk_bf16_f32', body)):4d}")
