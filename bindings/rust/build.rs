fn main() {
    // librtw_hip.so is produced by `make -C raytracing-in-a-weekend_amd/csrc`
    println!("cargo:rustc-link-search=native=../../raytracing-in-a-weekend_amd");
    println!("cargo:rustc-link-lib=dylib=rtw_hip");
}
