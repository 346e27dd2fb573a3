// Link against libmgym.so.  MGYM_LIB_DIR = directory holding the library
// (in this repository: modurl_gym_amd/, produced by `make -C modurl_gym_amd/csrc`).
fn main() {
    if let Ok(dir) = std::env::var("MGYM_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=mgym");
    println!("cargo:rerun-if-env-changed=MGYM_LIB_DIR");
}
