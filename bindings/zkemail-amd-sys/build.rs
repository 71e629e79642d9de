// Link against the prebuilt engine.  libzkemail_amd.so is built by `python -m zkemail_rs_amd.build` (hipcc,
// --offload-arch=gfx950) in the repository root; point ZKEMAIL_AMD_LIB_DIR at the directory that holds it
// (default: <repo>/zkemail.rs_amd).  The HIP runtime comes with it (/opt/rocm/lib, or ROCM_PATH).
use std::{env, path::PathBuf};

fn main() {
    println!("cargo:rerun-if-env-changed=ZKEMAIL_AMD_LIB_DIR");
    println!("cargo:rerun-if-env-changed=ROCM_PATH");
    let manifest = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap());
    let default_dir = manifest.join("..").join("..").join("zkemail.rs_amd");
    let lib_dir = env::var("ZKEMAIL_AMD_LIB_DIR").map(PathBuf::from).unwrap_or(default_dir);
    let rocm = env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".to_string());
    println!("cargo:rustc-link-search=native={}", lib_dir.display());
    println!("cargo:rustc-link-search=native={}/lib", rocm);
    println!("cargo:rustc-link-lib=dylib=zkemail_amd");
    println!("cargo:rustc-link-lib=dylib=amdhip64");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", lib_dir.display());
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}/lib", rocm);
    // re-exported to dependents as DEP_ZKEMAIL_AMD_INCLUDE
    println!("cargo:include={}", manifest.join("..").join("..").join("include").display());
}
