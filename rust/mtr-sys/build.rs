fn main() {
    // libmtr.so is built by mt_renderer_amd/csrc/Makefile (hipcc, gfx950)
    let dir = std::env::var("MTR_LIB_DIR").unwrap_or_else(|_| "../../mt_renderer_amd".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=mtr");
}
