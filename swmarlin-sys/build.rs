// Links libswmarlin.so.  SWMARLIN_LIB_DIR points at the directory that holds it (simpleworks_amd/ in the library's
// repository after `make -C simpleworks_amd/csrc`, i.e. hipcc --offload-arch=gfx950); the HIP runtime comes from ROCm.
use std::env;

fn main() {
    // the GPU-free pin kit (tests/pin_golden.rs) uses none of the library: nothing to link
    if env::var_os("CARGO_FEATURE_PIN").is_some() {
        return;
    }
    println!("cargo:rerun-if-env-changed=SWMARLIN_LIB_DIR");
    println!("cargo:rerun-if-env-changed=ROCM_PATH");
    if let Ok(dir) = env::var("SWMARLIN_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    }
    let rocm = env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".to_string());
    println!("cargo:rustc-link-search=native={}/lib", rocm);
    println!("cargo:rustc-link-lib=dylib=swmarlin");
    println!("cargo:rustc-link-lib=dylib=amdhip64");
}
