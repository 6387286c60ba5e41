// Builds libcaustic_hip.so with hipcc for gfx950 and links it.
// CAUSTIC_HIP_SRC points at the directory holding caustic_hip.hip (rust-raytracer_amd/csrc of this repository);
// ROCM_PATH (default /opt/rocm) locates hipcc.  The flags are the ones csrc/Makefile uses: -ffp-contract=off is
// required (the device code reproduces caustic_core's f64 expressions operation by operation).
use std::{env, path::PathBuf, process::Command};

fn main() {
    let src = PathBuf::from(env::var("CAUSTIC_HIP_SRC").expect("set CAUSTIC_HIP_SRC to .../rust-raytracer_amd/csrc"));
    let out = PathBuf::from(env::var("OUT_DIR").unwrap());
    let rocm = env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".into());
    let lib = out.join("libcaustic_hip.so");
    let status = Command::new(format!("{rocm}/bin/hipcc"))
        .args(["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-o"])
        .arg(&lib)
        .arg(src.join("caustic_hip.hip"))
        .status()
        .expect("hipcc not found");
    assert!(status.success(), "hipcc failed");
    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=caustic_hip");
    for f in ["caustic_hip.hip", "kernels.hpp", "scene_host.hpp", "device_scene.h"] {
        println!("cargo:rerun-if-changed={}", src.join(f).display());
    }
}
