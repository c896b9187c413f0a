// Links libknaster_hip.so (built by `python -m knaster_amd.build`, hipcc --offload-arch=gfx950).
// KNASTER_HIP_LIB_DIR overrides the default location inside this repository.
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var("KNASTER_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../../knaster_amd/csrc")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=knaster_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=KNASTER_HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=../../../include/knaster_hip.h");
}
