// Links libmi355rt.so.  MI355RT_LIB_DIR names the directory that holds it (ray_tracer_amd/ of the library's tree
// after `make -C ray_tracer_amd/csrc`); the ROCm runtime it depends on is found through the usual loader paths.
fn main() {
    if let Ok(dir) = std::env::var("MI355RT_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    }
    println!("cargo:rustc-link-lib=dylib=mi355rt");
    println!("cargo:rerun-if-env-changed=MI355RT_LIB_DIR");
}
