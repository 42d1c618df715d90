// Source-only (see ../README.md).  Links the prebuilt libzkp_hip.so; ZKP_HIP_LIB_DIR points at the directory that holds it
// (zkp-implementation_amd/ after `python zkp-implementation_amd/build.py`).
fn main() {
    let dir = std::env::var("ZKP_HIP_LIB_DIR").unwrap_or_else(|_| "../../zkp-implementation_amd".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=zkp_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=ZKP_HIP_LIB_DIR");
}
