// Link against libzkt_hip.so.  ZKT_LIB_DIR = the directory holding it (default: ../../zk-toolkit_amd of this checkout).
fn main() {
    let dir = std::env::var("ZKT_LIB_DIR").unwrap_or_else(|_| format!("{}/../../zk-toolkit_amd", env!("CARGO_MANIFEST_DIR")));
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=zkt_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=ZKT_LIB_DIR");
}
