// placeholder, replaced below
int main() { return 0; }
