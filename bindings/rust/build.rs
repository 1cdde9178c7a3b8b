// Links libimt_hip.so.  IMT_HIP_LIB_DIR = the directory holding it (in this repository:
// indexed-merkle-tree-halo2_amd/csrc after `make`); the HIP runtime it depends on comes from /opt/rocm/lib.
fn main() {
    let dir = std::env::var("IMT_HIP_LIB_DIR").unwrap_or_else(|_| "../../indexed-merkle-tree-halo2_amd/csrc".into());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=imt_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=IMT_HIP_LIB_DIR");
}
