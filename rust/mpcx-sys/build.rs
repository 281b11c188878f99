// Builds libmpcx.so from the HIP sources of this repository with hipcc for gfx950 and links it.
// MPCX_LIB_DIR (a directory that already holds libmpcx.so, e.g. mpc-rs_amd/csrc after `make`) skips the build.
use std::env;
use std::path::PathBuf;
use std::process::Command;

// the SRCS list of mpc-rs_amd/csrc/Makefile, in its order (tests/test_host_cpu.py holds the two against each other)
const SOURCES: [&str; 14] = [
    "mpcx_api.hip", "qp_lane.hip", "qp_wg.hip", "qp_stream.hip", "hessian_mfma.hip", "dense_w4.hip", "solve.hip", "solve_large.hip",
    "gain_rows.hip", "plants.hip", "closed_loop.hip", "rollout.hip", "util.hip", "comm.hip",
];

fn main() {
    println!("cargo:rerun-if-env-changed=MPCX_LIB_DIR");
    println!("cargo:rerun-if-env-changed=HIPCC");
    if let Ok(dir) = env::var("MPCX_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-lib=dylib=mpcx");
        return;
    }
    let root = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../..");
    let csrc = root.join("mpc-rs_amd/csrc");
    let include = root.join("include");
    let out = PathBuf::from(env::var("OUT_DIR").unwrap());
    let hipcc = env::var("HIPCC").unwrap_or_else(|_| "/opt/rocm/bin/hipcc".to_string());
    let mut cmd = Command::new(hipcc);
    cmd.args(["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"]);
    // MFMA accumulators stay in ordinary VGPRs (see mpc-rs_amd/csrc/Makefile)
    cmd.args(["-mllvm", "-amdgpu-mfma-vgpr-form=1"]);
    cmd.arg(format!("-I{}", include.display()));
    for s in SOURCES {
        let path = csrc.join(s);
        println!("cargo:rerun-if-changed={}", path.display());
        cmd.arg(path);
    }
    println!("cargo:rerun-if-changed={}", include.join("mpcx.h").display());
    cmd.arg("-o").arg(out.join("libmpcx.so"));
    cmd.arg("-ldl"); // comm.hip binds RCCL with dlopen at the first communicator call
    let status = cmd.status().expect("hipcc could not be started (set HIPCC, or MPCX_LIB_DIR to a prebuilt libmpcx.so)");
    assert!(status.success(), "hipcc failed to build libmpcx.so");
    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=mpcx");
}
